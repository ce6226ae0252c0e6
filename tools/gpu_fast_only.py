import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
B = 256
imgs = np.concatenate([pkg.synth.synth_batch(0, 32)] * 8)
ex = pkg.Extractor()
res = ex.extract_batch(imgs)
res = ex.extract_batch(imgs)
print(os.environ.get("AOS2_LIB", "current").split("/")[-1], {k: round(v, 3) for k, v in ex.last_timing().items()}, "fast_ms %.3f" % ex.bench_fast(10))

import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
imgs = np.concatenate([pkg.synth.synth_batch(0, 32)] * 8)
ex = pkg.Extractor()
ex.set_chunks(1)
ex.extract_batch(imgs); ex.extract_batch(imgs)
print(os.environ.get("AOS2_LIB", "current").split("/")[-1], "describe_ms %.4f" % ex.bench_describe(20), "fast_ms %.4f" % ex.bench_fast(20))

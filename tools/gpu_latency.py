import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
img = pkg.synth.synth_image(5)
ex = pkg.Extractor()
ex(img)
ts = []
for i in range(30):
    t = time.perf_counter(); ex(img); ts.append(time.perf_counter() - t)
print("single frame host API: median %.3f ms" % (np.median(ts) * 1e3), ex.last_timing())
for B in (16, 256):
    imgs = pkg.synth.synth_batch(0, min(B, 32)); imgs = np.concatenate([imgs] * (B // len(imgs)))
    ex.extract_batch(imgs)
    t = time.perf_counter()
    for i in range(5): ex.extract_batch(imgs)
    dt = (time.perf_counter() - t) / 5
    print("host API batch %d: %.2f ms -> %.0f frames/s (python unpack of results included)" % (B, dt * 1e3, B / dt))

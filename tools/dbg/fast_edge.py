import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
orc = g.load_oracle()
def same(a, b):
    return len(a[0]) == len(b[0]) and np.array_equal(np.asarray(a[0]).view(np.uint8), np.asarray(b[0]).view(np.uint8)) and np.array_equal(a[1], b[1])
ex = pkg.Extractor()
cases = {}
low = pkg.synth.synth_image(9); cases["low"] = (low.astype(np.int32) // 6 + 100).astype(np.uint8)
cases["sat"] = np.where(pkg.synth.synth_image(10) > 128, 255, 0).astype(np.uint8)
rng = np.random.default_rng(7)
cases["noise"] = (rng.integers(0, 2, (480, 640)) * 255).astype(np.uint8)
yy, xx = np.mgrid[0:480, 0:640]
cases["checker"] = (((xx // 3) + (yy // 3)) % 2 * 200 + 20).astype(np.uint8)
cases["plain"] = pkg.synth.synth_image(3)
for n, im in cases.items():
    a = ex(im); b = orc.Extractor().extract(im)
    print(n, len(a[0]), len(b[0]), same(a, b))
    if not same(a, b):
        ca = ex.debug_candidates(im) if hasattr(ex, "debug_candidates") else None
big = np.zeros((333, 700), np.uint8)
big[:, :517] = pkg.synth.synth_image(11, 517, 333)
view = big[:, :517]
ex2, oe2 = pkg.Extractor(nfeatures=700), orc.Extractor(nfeatures=700)
a, b = ex2(view), oe2.extract(np.ascontiguousarray(view))
print("view", len(a[0]), len(b[0]), same(a, b))
if not same(a, b):
    ka, kb = np.asarray(a[0]), np.asarray(b[0])
    n = min(len(ka), len(kb))
    bad = [i for i in range(n) if ka[i].tobytes() != kb[i].tobytes()]
    print(len(bad), bad[:10]); print(ka[bad[:3]]); print(kb[bad[:3]])
ex3 = pkg.Extractor(nfeatures=800, scale_factor=1.5, nlevels=4, ini_th=30, min_th=10)
oe3 = orc.Extractor(nfeatures=800, scale_factor=1.5, nlevels=4, ini_th=30, min_th=10)
img = pkg.synth.synth_image(12)
a, b = ex3(img), oe3.extract(img)
print("ex3", len(a[0]), len(b[0]), same(a, b))
if not same(a, b):
    ka, kb = np.asarray(a[0]), np.asarray(b[0])
    n = min(len(ka), len(kb))
    bad = [i for i in range(n) if ka[i].tobytes() != kb[i].tobytes()]
    print(len(bad), bad[:10]); print(ka[bad[:3]]); print(kb[bad[:3]])

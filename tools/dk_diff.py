"""Which keypoints / descriptor rows of the HIP extractor differ from the oracle's on one synthetic frame (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
c = pkg.synth.CONFIGS["tum"]
ex = pkg.Extractor(nfeatures=c["nfeatures"]); oe = O.Extractor(nfeatures=c["nfeatures"])
img = pkg.synth.synth_image(1, c["w"], c["h"])
k, d = ex(img); k2, d2 = oe.extract(img)
print(len(k), len(k2))
n = min(len(k), len(k2))
for f in k.dtype.names:
    bad = np.nonzero(k[f][:n] != k2[f][:n])[0]
    print(f, len(bad), bad[:40])
bad = np.nonzero((d[:n] != d2[:n]).any(1))[0]
print("desc rows", len(bad), bad[:60])
if len(bad):
    i = bad[0]; print(i, d[i], d2[i], k[i], k2[i])

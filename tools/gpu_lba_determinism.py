"""Stress check of the last-workgroup hand-over in the LocalBA landmark kernels (device-scope write-through stores + relaxed
counter instead of __threadfence): N solves of the same batch -- alone and with a second handle solving concurrently and
the extractor running on another stream -- must give bit-identical results every time (a lost update would change a
chi2 sum, an LM decision, and so the poses)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
u = [pkg.synth.synth_lba_problem(i, n_points=8000) for i in range(4)] + [pkg.synth.synth_lba_problem(40 + i) for i in range(4)]
u += pkg.synth.synth_lba_problems(pkg.synth.lba_window_mix(5, 8))   # round 4: windows of different sizes, reduced systems inside and beyond LDS
probs = [u[i % len(u)] for i in range(32)]
def key(res):
    return b"".join(np.asarray(r["pose_Tcw"]).tobytes() + np.asarray(r["point_xyz"]).tobytes() + np.asarray(r["edge_outlier"]).tobytes() for r in res)
ba = pkg.LocalBA()
ref = key(ba.LocalBundleAdjustmentBatch(probs))
bad = 0
for layout in ("walk", "slots"):
    os.environ["AOS2_LBA_LAYOUT"] = layout
    for i in range(N // 2):
        bad += key(ba.LocalBundleAdjustmentBatch(probs)) != ref
print("sequential: %d solves, %d differ" % (2 * (N // 2), bad), flush=True)
del os.environ["AOS2_LBA_LAYOUT"]
# concurrently: a second LocalBA handle and the extractor keep the device busy
stop = False
def other_lba():
    b2 = pkg.LocalBA()
    while not stop:
        b2.LocalBundleAdjustmentBatch(probs[:16])
def extractor():
    ex = pkg.Extractor()
    imgs = pkg.synth.synth_batch(0, 32)
    while not stop:
        ex.extract_batch(imgs)
th = [threading.Thread(target=other_lba), threading.Thread(target=extractor)]
[t.start() for t in th]
bad2 = 0
for i in range(N):
    bad2 += key(ba.LocalBundleAdjustmentBatch(probs)) != ref
stop = True
[t.join() for t in th]
print("with a concurrent LocalBA handle and extractor: %d solves, %d differ" % (N, bad2), flush=True)
sys.exit(1 if bad or bad2 else 0)

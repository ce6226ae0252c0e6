"""The keyframe legs of bench.py's step alone (64 keyframes x 10 neighbours: SearchForTriangulation + Fuse both ways on device-resident
keyframes), a few repetitions -- run under `rocprofv3 --kernel-trace --stats` by tools/prof_r03.sh.  Prints the wall time of the calls."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
scen = pkg.scenario.tracking_scenario(100, 64, n_unique=32)
tc = pkg.chain.TrackingChain(scen, n_local=1500)
voc = pkg.synth.synth_vocabulary(400, 10, 6)
kw = pkg.chain.KeyFrameWork(tc, voc, 64, n_nb=10)
kw.run()
for _ in range(5):
    t0 = time.perf_counter()
    kw.run()
    print("keyframe work: %.3f ms wall (SearchForTriangulation %.3f, Fuse x2 %.3f); %d pairs, matches per pair %.1f, fused per pair %.1f, reverse %.1f"
          % ((time.perf_counter() - t0) * 1e3, kw.last_ms[0], kw.last_ms[1], len(kw.kf1), kw.nm.mean(), (kw.best_idx >= 0).sum(1).mean(),
             (kw.rev_idx >= 0).sum(1).mean()))

"""Independent LocalBA windows solved at the same time, one handle + one host thread each (bench.py's
extra.local_ba.concurrent_windows): wall time and windows/s for 1, 2, 4, 8 windows."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
NW = 8
bas = [pkg.LocalBA() for _ in range(NW)]
probs = [pkg.synth.synth_lba_problem(i) for i in range(NW)]
for b, q in zip(bas, probs):
    b.LocalBundleAdjustment(q)
for n in (1, 2, 4, 8):
    best = 1e9
    for rep in range(3):
        ths = [threading.Thread(target=b.LocalBundleAdjustment, args=(q,)) for b, q in zip(bas[:n], probs[:n])]
        t = time.perf_counter()
        [x.start() for x in ths]
        [x.join() for x in ths]
        best = min(best, time.perf_counter() - t)
    print("%d windows: %.2f ms wall, %.0f windows/s" % (n, best * 1e3, n / best))

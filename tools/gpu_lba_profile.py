import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
prob = pkg.synth.synth_lba_problem(0)
ba = pkg.LocalBA()
for _ in range(3):
    t0 = time.time(); r = ba.LocalBundleAdjustment(prob); print("wall %.2f ms dev %.2f ms" % ((time.time() - t0) * 1e3, r["ms_device"]), r["iters"])

"""LocalBA timing on the GPU box: single window (12 k and 24 k edges), batches of independent windows."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
ba = pkg.LocalBA()
for name, prob in (("12k", pkg.synth.synth_lba_problem(0)), ("24k", pkg.synth.synth_lba_problem(0, n_points=8000))):
    print(name, "edges", prob["n_edges"], "points", prob["n_points"], "poses", prob["n_poses"])
    for _ in range(4):
        t0 = time.time(); r = ba.LocalBundleAdjustment(prob)
        print("  wall %.2f ms dev %.2f ms" % ((time.time() - t0) * 1e3, r["ms_device"]), r["iters"], r["trials"])
if "--batch" in sys.argv:
    for n in (2, 8, 32):
        probs = [pkg.synth.synth_lba_problem(i, n_points=8000) for i in range(n)]
        ba.LocalBundleAdjustmentBatch(probs)
        t0 = time.time(); rs = ba.LocalBundleAdjustmentBatch(probs); w = (time.time() - t0) * 1e3
        print("batch of %d 24k-edge windows: wall %.2f ms dev %.2f ms = %.0f windows/s" % (n, w, rs[0]["ms_device"], n / w * 1e3),
              [r["trials"] for r in rs[:4]])

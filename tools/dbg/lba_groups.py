import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
cfgs = [dict(seed=61, n_local=1, n_fixed=4, n_points=120), dict(seed=62, n_local=6, n_fixed=2, n_points=260),
        dict(seed=63, n_local=12, n_fixed=3, n_points=330, stereo_frac=0.4), dict(seed=64, n_local=3, n_fixed=0, n_points=90, include_kf0=True)]
uniq = [pkg.synth.synth_lba_problem(**c) for c in cfgs]
for i, p in enumerate(uniq):
    print("problem", i, "poses", p["n_poses"], "points", p["n_points"], "edges", p["n_edges"], flush=True)
    r = pkg.LocalBA().LocalBundleAdjustment(p)
    print("  alone ok", r["status"], r["iters"], flush=True)
for layout in ("slots", "walk"):
    os.environ["AOS2_LBA_LAYOUT"] = layout
    ba = pkg.LocalBA()
    for n in (2, 7, 8, 9, 16, 17, 33):
        idx = [(3 * i + n) % len(uniq) for i in range(n)]
        print(layout, n, flush=True)
        got = ba.LocalBundleAdjustmentBatch([uniq[j] for j in idx])
        print("  ok", [x["status"] for x in got][:4], flush=True)

import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
cfgs = [dict(seed=61, n_local=1, n_fixed=4, n_points=120), dict(seed=62, n_local=6, n_fixed=2, n_points=260),
        dict(seed=63, n_local=12, n_fixed=3, n_points=330, stereo_frac=0.4), dict(seed=64, n_local=3, n_fixed=0, n_points=90, include_kf0=True)]
uniq = [pkg.synth.synth_lba_problem(**c) for c in cfgs]
alone = [pkg.LocalBA().LocalBundleAdjustment(p) for p in uniq]
def diff(a, b):
    out = []
    for k in ("pose_Tcw", "point_xyz", "edge_outlier", "edge_chi2"):
        if a[k].tobytes() != b[k].tobytes(): out.append((k, float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max())))
    for k in ("iters", "trials", "status"):
        if a[k] != b[k]: out.append((k, a[k], b[k]))
    return out
for layout in ("slots", "walk"):
    os.environ["AOS2_LBA_LAYOUT"] = layout
    ba = pkg.LocalBA()
    for n in (16, 17, 33):
        for rep in range(int(os.environ.get("REPS", "40"))):
            idx = [(3 * i + n) % len(uniq) for i in range(n)]
            got = ba.LocalBundleAdjustmentBatch([uniq[j] for j in idx])
            bad = [(i, j, diff(gg, alone[j])) for i, (gg, j) in enumerate(zip(got, idx)) if diff(gg, alone[j])]
            if bad: print(layout, n, rep, "bad windows:", bad[:4], len(bad), flush=True)
    print(layout, "done", flush=True)

"""Where do the device solver and the oracle part on a window that starts off the optimum?  Both are stopped after k iterations
(iters = (k, 0), then (5, k)); the double-precision chi2 / lambda of the two and the float32 write-back are compared per k."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
S = pkg.synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import lba_sensitivity as LS

names = os.environ.get("PROBE", "").split(",") if os.environ.get("PROBE") else None
cands = LS.hard_test_problems() + LS.bench_problems(int(os.environ.get("LBA_SENS_WINDOWS", "64")))
ba = pkg.LocalBA()
for name, prob in cands:
    if not ("HARD" in name or "_hard_problem" in name):
        continue
    if names and not any(n in name for n in names):
        continue
    full_o = O.lba_solve(prob)
    full_g = ba.LocalBundleAdjustment(prob)
    dp = float(np.abs(full_g["pose_Tcw"] - full_o["pose_Tcw"]).max()); dx = float(np.abs(full_g["point_xyz"] - full_o["point_xyz"]).max())
    print("\n%s: full run poses %.2e points %.2e; oracle iters %s trials %d, device iters %s trials %s" % (name, dp, dx, full_o["iters"], full_o["trials"], full_g["iters"], full_g["trials"]), flush=True)
    if os.environ.get("PROBE_FULL_ONLY"):
        continue
    for it in [(k, 0) for k in range(1, 6)] + [(5, k) for k in range(1, 11)]:
        o = O.lba_solve(prob, iters1=it[0], iters2=it[1])
        d = ba.LocalBundleAdjustment(prob, iters=it)
        oc, ol = o["chi2_trace"][-1], o["lambda_trace"][-1]
        print("   iters %-8s oracle chi2 %.10e lambda %.6e trials %2d | device rel chi2 %.2e rel lambda %.2e trials %s | poses %.2e points %.2e outliers %d" %
              (it, oc, ol, o["trials"], abs(d["final_chi2"] - oc) / max(abs(oc), 1e-300), abs(d["final_lambda"] - ol) / max(abs(ol), 1e-300), d["trials"],
               float(np.abs(d["pose_Tcw"] - o["pose_Tcw"]).max()), float(np.abs(d["point_xyz"] - o["point_xyz"]).max()), int((d["edge_outlier"] != o["edge_outlier"]).sum())), flush=True)

import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
prob = pkg.synth.synth_lba_problem(6, n_local=5, n_fixed=3, n_points=250)
bad = dict(prob)
rng = np.random.default_rng(3)
obs = prob["edge_obs"].copy()
obs[:, :2] += rng.choice([-1.0, 1.0], (len(obs), 2)).astype(np.float32) * rng.uniform(60, 200, (len(obs), 2)).astype(np.float32)
bad["edge_obs"] = obs
want = O.lba_solve(bad)
print("oracle iters", want["iters"], "trials", want["trials"], "outliers", int(want["edge_outlier"].sum()), "of", len(obs))
for k in ("chi2_trace", "lambda_trace"):
    if k in want: print(k, [float("%.9e" % v) for v in want[k]])
got = pkg.LocalBA().LocalBundleAdjustment(bad)
print("device iters", got["iters"], "trials", got["trials"], "outliers", int(got["edge_outlier"].sum()))
print("max dpose", np.abs(got["pose_Tcw"] - want["pose_Tcw"]).max(), "max dpoint", np.abs(got["point_xyz"] - want["point_xyz"]).max())

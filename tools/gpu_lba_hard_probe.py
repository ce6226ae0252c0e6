"""Windows that start off the optimum (synth.perturb_lba_problem) through the device solver and the oracle: trials, worst differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
mix = pkg.synth.lba_window_mix(0, 64, hard_every=8)
ba = pkg.LocalBA()
for hard in [(0.5, 3, 2), (0.5, 3, 0.0), (0.3, 1.0, 0.5), (0.5, 1.0, 0.0), (0.5, 3, 0.5)]:
    probs = []
    for i in range(7, 64, 8):
        m = dict(mix[i]); m["hard"] = hard; m["stereo_frac"] = 0.3
        probs.append(pkg.synth._lba_from_kwargs(m))
    got = ba.LocalBundleAdjustmentBatch(probs)
    rows = []
    for p, gt in zip(probs, got):
        w = O.lba_solve(p)
        rows.append((sum(gt["trials"]), w["trials"], float(np.abs(gt["pose_Tcw"] - w["pose_Tcw"]).max()), float(np.abs(gt["point_xyz"] - w["point_xyz"]).max()),
                     int((gt["edge_outlier"] != w["edge_outlier"]).sum())))
    print(hard, "rounds", ba.last_program(), "trials(dev,oracle) / dpose / dpoint / outlier diffs:", [(a, b, "%.1e" % c, "%.1e" % d, e) for a, b, c, d, e in rows], flush=True)

import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g, numpy as np
pkg=g.load_package(); O=g.load_oracle()
ba=pkg.LocalBA()
rng=np.random.default_rng(5)
for n in (3,4,5,6,7):
    worst=0; bad=0; cnt=0; big=[]
    for seed in range(400):
        p=pkg.synth.synth_pose_problem(3000+seed, n=n, stereo_frac=float(rng.choice([0.0,0.5,1.0])), outlier_frac=float(rng.choice([0.0,0.1,0.4])), cfg=("kitti","tum")[seed%2])
        w=O.pose_optimization(p); r=ba.PoseOptimization(p)
        d=float(np.abs(r["Tcw"].astype(np.float64)-w["Tcw"].astype(np.float64)).max())
        if not np.isfinite(d): d=1e9
        worst=max(worst,d); cnt+=1
        if d>1e-5: big.append((seed,d))
        bad += int((r["outlier"]!=w["outlier"]).any() or r["n_inliers"]!=w["n_inliers"])
    print(f"n={n}: worst |dT| {worst:.3e}, >1e-5: {len(big)}/{cnt}, outlier mismatches {bad}", big[:6])

"""One-off parity sweep of the remaining rows (a time-boxed slice of it runs in tests/test_fuzz_gpu.py): stereo matching, vocabulary transform,
PoseOptimization and LocalBundleAdjustment with random sizes / parameters vs the oracle.
python tools/gpu_fuzz_rest.py [n_cases [seconds]]"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle(); S = pkg.synth
budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 1e18   # optional time budget in seconds (tests/test_fuzz_gpu.py)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(4321)
bad = 0
t0 = time.time()


def close(a, b, tol=1e-5):
    return (np.abs(a.astype(np.float64) - b.astype(np.float64)) <= tol + 2 * np.spacing(np.abs(b).astype(np.float32))).all()


ran = 0
for c in range(n_cases):
    if time.time() - t0 > budget_s:
        break
    ran += 1
    # ---- stereo: random image size / feature count
    w, h = int(rng.integers(200, 1300)), int(rng.integers(160, 720))
    h = min(h, int(1.6 * w))
    nf = int(rng.choice([200, 1000, 2000]))
    left, right, _ = S.synth_stereo_pair(300 + c, w, h, max_disp=int(rng.choice([16, 48, 96])))
    mbf = np.float32(rng.choice([40.0, 386.1])); mb = np.float32(mbf / np.float32(rng.choice([435.2, 718.856])))
    try:
        eL, eR = O.Extractor(nfeatures=nf), O.Extractor(nfeatures=nf)
        kl, dl = eL.extract(left); kr, dr = eR.extract(right)
        our, odp, on = O.compute_stereo_matches(eL, eR, kl, dl, kr, dr, mb, mbf)
        xl, xr = pkg.Extractor(nfeatures=nf), pkg.Extractor(nfeatures=nf)
        gkl, gdl = xl(left); gkr, gdr = xr(right)
        ur, dp = pkg.ComputeStereoMatches(xl, xr, gkl, gdl, gkr, gdr, mb, mbf)
        if ur.tobytes() != our.tobytes() or dp.tobytes() != odp.tobytes() or gkl.tobytes() != kl.tobytes():
            bad += 1; print("MISMATCH stereo", c, (w, h, nf))
    except Exception as e:
        if "too small" not in repr(e) and "twice as tall" not in repr(e) and "failed: -3" not in repr(e) and "failed: -4" not in repr(e):
            bad += 1; print("ERROR stereo", c, (w, h, nf), repr(e))
    # ---- PoseOptimization
    n = int(rng.choice([1, 2, 3, 4, 5, 6, 9, 64, 257, 800, 1025, 3000]))
    p = S.synth_pose_problem(700 + c, n=n, stereo_frac=float(rng.choice([0.0, 0.5, 1.0])), outlier_frac=float(rng.choice([0.0, 0.1, 0.4])),
                             cfg=("kitti", "tum")[c % 2])
    wv, gv = O.pose_optimization(p), pkg.LocalBA().PoseOptimization(p)
    if gv["n_inliers"] != wv["n_inliers"] or not (gv["outlier"] == wv["outlier"]).all() or not close(gv["Tcw"].reshape(1, 16), wv["Tcw"].reshape(1, 16)):
        bad += 1; print("MISMATCH pose", c, n)
    # ---- LocalBA (every 3rd case: the oracle needs ~10-100 ms)
    if c % 3 == 0:
        cfg = dict(seed=900 + c, n_local=int(rng.integers(1, 12)), n_fixed=int(rng.integers(0, 8)), n_points=int(rng.integers(20, 900)),
                   stereo_frac=float(rng.choice([0.0, 0.5, 1.0])))
        try:
            prob = S.synth_lba_problem(**cfg)
            if prob["n_edges"] == 0 or prob["n_points"] == 0:   # nothing to optimise: rejected as a bad problem by design
                continue
            wv, gv = O.lba_solve(prob), pkg.LocalBA().LocalBundleAdjustment(prob)
            if gv["iters"] != wv["iters"] or not close(gv["pose_Tcw"], wv["pose_Tcw"]) or not close(gv["point_xyz"], wv["point_xyz"]) \
               or not (gv["edge_outlier"] == wv["edge_outlier"]).all():
                bad += 1; print("MISMATCH lba", c, cfg)
        except Exception as e:
            bad += 1; print("ERROR lba", c, cfg, repr(e))
print("cases", ran, "of", n_cases, "MISMATCHES / ERRORS", bad, "time %.1f s" % (time.time() - t0))

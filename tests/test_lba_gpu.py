"""HIP LocalBundleAdjustment vs the oracle.  Floating point (IEEE double inside, float32 at the
boundary like the reference): tolerance 1e-5 absolute on the float32 poses/points written back
(north_star), checked on top of the float32 quantisation of the stored value."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-5


def close(a, b, tol=TOL):
    # 1e-5 absolute, plus one float32 ulp of the magnitude (values ~40 m have ulp 3.8e-6)
    return (np.abs(a.astype(np.float64) - b.astype(np.float64)) <= tol + 2 * np.spacing(np.abs(b).astype(np.float32))).all()


@pytest.mark.parametrize("cfg", [dict(seed=1, n_local=3, n_fixed=2, n_points=60, stereo_frac=0.5),
                                 dict(seed=2, n_local=6, n_fixed=4, n_points=400, stereo_frac=0.0),
                                 dict(seed=4, n_local=5, n_fixed=0, n_points=300, include_kf0=True),
                                 dict(seed=0), dict(seed=3, include_kf0=True, outlier_frac=0.15)])
def test_lba_vs_oracle(pkg, oracle, gpu, cfg):
    prob = pkg.synth.synth_lba_problem(**cfg)
    want = oracle.lba_solve(prob)
    got = pkg.LocalBA().LocalBundleAdjustment(prob)
    assert got["status"] == 0 and got["iters"] == want["iters"]
    assert close(got["pose_Tcw"], want["pose_Tcw"])
    assert close(got["point_xyz"], want["point_xyz"])
    assert (got["edge_outlier"] == want["edge_outlier"]).all()
    assert abs(got["final_chi2"] - want["chi2_trace"][-1]) <= 1e-6 * want["chi2_trace"][-1]
    assert np.allclose(got["edge_chi2"], want["edge_chi2"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["lba_3kf", "lba_14kf"])
def test_lba_golden(pkg, gpu, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    prob = {k: g[k] for k in g.files if not k.startswith("out_")}
    for k in ("n_poses", "n_points", "n_edges"):
        prob[k] = int(prob[k])
    got = pkg.LocalBA().LocalBundleAdjustment(prob)
    assert close(got["pose_Tcw"], g["out_pose_Tcw"]) and close(got["point_xyz"], g["out_point_xyz"])
    assert (got["edge_outlier"] == g["out_outlier"]).all()


def test_lba_stop_flag_and_properties(pkg, oracle, gpu):
    prob = pkg.synth.synth_lba_problem(5, n_local=4, n_fixed=3, n_points=200)
    ba = pkg.LocalBA()
    flag = np.ones(1, np.uint8)
    r = ba.LocalBundleAdjustment(prob, stop_flag=flag)
    assert r["status"] == pkg.capi.AOS2_ERR_STOPPED and r["iters"] == (0, 0)
    assert (r["pose_Tcw"] == prob["pose_Tcw"]).all() and (r["point_xyz"] == prob["point_xyz"]).all()
    r = ba.LocalBundleAdjustment(prob)
    fixed = prob["pose_fixed"].astype(bool)
    assert np.abs(r["pose_Tcw"][fixed] - prob["pose_Tcw"][fixed]).max() < 1e-6   # fixed cameras do not move
    # a solved problem stays solved (idempotence up to the LM stop rule)
    p2 = dict(prob)
    p2["pose_Tcw"], p2["point_xyz"] = r["pose_Tcw"], r["point_xyz"]
    r2 = ba.LocalBundleAdjustment(p2)
    assert np.abs(r2["pose_Tcw"] - r["pose_Tcw"]).max() < 5e-2
    # iteration counts: only the first pass
    r3 = ba.LocalBundleAdjustment(prob, iters=(5, 0))
    w3 = oracle.lba_solve(prob, iters1=5, iters2=0)
    assert close(r3["pose_Tcw"], w3["pose_Tcw"]) and close(r3["point_xyz"], w3["point_xyz"])


# ---- Optimizer::PoseOptimization -----------------------------------------
@pytest.mark.parametrize("cfg", [dict(seed=0), dict(seed=1, stereo_frac=0.0, cfg="tum"), dict(seed=2, n=300, outlier_frac=0.3),
                                 dict(seed=3, n=2000), dict(seed=4, n=8), dict(seed=5, n=2), dict(seed=6, n=40, rot_err=0.05)])
def test_pose_optimization_vs_oracle(pkg, oracle, gpu, cfg):
    prob = pkg.synth.synth_pose_problem(**cfg)
    want = oracle.pose_optimization(prob)
    got = pkg.LocalBA().PoseOptimization(prob)
    assert got["n_inliers"] == want["n_inliers"] and got["n_bad"] == want["n_bad"]
    assert (got["outlier"] == want["outlier"]).all()
    assert close(got["Tcw"].reshape(1, 16), want["Tcw"].reshape(1, 16))


def test_pose_optimization_fewer_than_three_correspondences(pkg, oracle, gpu):
    """nInitialCorrespondences < 3 (:355-356): pose unchanged, 0 inliers, and mvbOutlier all false -- also right after a
    call on the same handle that left outlier flags in the device arena (found by tools/gpu_fuzz_rest.py)."""
    ba = pkg.LocalBA()
    big = pkg.synth.synth_pose_problem(1, n=900, outlier_frac=0.5)
    assert ba.PoseOptimization(big)["outlier"].any()
    for seed in range(700, 720):
        for n in (1, 2):
            p = pkg.synth.synth_pose_problem(seed, n=n, stereo_frac=0.5)
            got, want = ba.PoseOptimization(p), oracle.pose_optimization(p)
            assert got["n_inliers"] == want["n_inliers"] == 0 and not got["outlier"].any() and not want["outlier"].any()
            assert got["Tcw"].tobytes() == p["Tcw"].astype(np.float32).tobytes()
        ba.PoseOptimization(big)


def test_pose_optimization_batch_and_golden(pkg, oracle, gpu):
    probs = [pkg.synth.synth_pose_problem(100 + i, n=600 + 37 * i) for i in range(24)]
    ba = pkg.LocalBA()
    res = ba.PoseOptimization(probs)
    for p, r in zip(probs, res):
        w = oracle.pose_optimization(p)
        assert r["n_inliers"] == w["n_inliers"] and (r["outlier"] == w["outlier"]).all()
        assert close(r["Tcw"].reshape(1, 16), w["Tcw"].reshape(1, 16))
    assert ba.pose_last_device_ms() > 0
    g = np.load(os.path.join(GOLD, "pose_500.npz"))
    prob = {k: g[k] for k in g.files if not k.startswith("out_")}
    prob["n"] = int(prob["n"])
    r = ba.PoseOptimization(prob)
    assert r["n_inliers"] == int(g["out_n_inliers"]) and (r["outlier"] == g["out_outlier"]).all()
    assert close(r["Tcw"].reshape(1, 16), g["out_Tcw"].reshape(1, 16))


def test_lba_concurrent_windows_from_threads(pkg, gpu):
    """Independent windows (one handle + one host thread each, SURVEY 8(e) "replicas only") solved at the same time
    give exactly the results of solving them one after the other: handles share nothing but the device."""
    import threading
    probs = [pkg.synth.synth_lba_problem(30 + i, n_local=4 + i, n_fixed=3, n_points=150 + 40 * i) for i in range(6)]
    serial = [pkg.LocalBA().LocalBundleAdjustment(q) for q in probs]
    handles = [pkg.LocalBA() for _ in probs]
    out = [None] * len(probs)

    def work(i):
        for _ in range(3):
            out[i] = handles[i].LocalBundleAdjustment(probs[i])

    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(probs))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for a, b in zip(out, serial):
        assert a is not None and a["status"] == 0 and a["iters"] == b["iters"]
        assert (a["pose_Tcw"] == b["pose_Tcw"]).all() and (a["point_xyz"] == b["point_xyz"]).all()
        assert (a["edge_outlier"] == b["edge_outlier"]).all()


def test_lba_schur_by_items_equals_lds_accumulation(pkg, oracle, gpu, monkeypatch):
    """The default Schur complement (one thread per (landmark, pose pair) item + one workgroup per block) and the former
    per-wave LDS accumulation (AOS2_SCHUR=partial) differ only in the summation order of f64 terms: same iterations,
    same outlier set, float32 results within the north_star tolerance of each other and of the oracle.  Includes a
    problem whose landmarks are partly seen by fixed keyframes only (no item stores their Dinv)."""
    for cfg in (dict(seed=11, n_local=7, n_fixed=5, n_points=500, stereo_frac=0.4), dict(seed=12, n_local=2, n_fixed=9, n_points=300),
                dict(seed=0)):
        prob = pkg.synth.synth_lba_problem(**cfg)
        monkeypatch.delenv("AOS2_SCHUR", raising=False)
        a = pkg.LocalBA().LocalBundleAdjustment(prob)
        monkeypatch.setenv("AOS2_SCHUR", "partial")
        b = pkg.LocalBA().LocalBundleAdjustment(prob)
        monkeypatch.delenv("AOS2_SCHUR", raising=False)
        want = oracle.lba_solve(prob)
        for got in (a, b):
            assert got["status"] == 0 and got["iters"] == want["iters"]
            assert close(got["pose_Tcw"], want["pose_Tcw"]) and close(got["point_xyz"], want["point_xyz"])
            assert (got["edge_outlier"] == want["edge_outlier"]).all()
        assert close(a["pose_Tcw"], b["pose_Tcw"]) and close(a["point_xyz"], b["point_xyz"])


def _hard_problem(pkg, seed, rot_sigma, trans_sigma, point_sigma):
    """a small window whose free keyframes and points start far from the optimum, so that Levenberg-Marquardt rejects
    steps (lambda grows, estimates are restored) and, for some seeds, gives up after ten failed trials"""
    def rot(axis, a):
        axis = axis / np.linalg.norm(axis)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
    prob = pkg.synth.synth_lba_problem(seed, n_local=5, n_fixed=3, n_points=250, stereo_frac=0.3)
    rng = np.random.default_rng(seed)
    T = prob["pose_Tcw"].copy().reshape(-1, 4, 4)
    for i in range(len(T)):
        if not prob["pose_fixed"][i]:
            R = rot(rng.normal(size=3), rot_sigma * rng.normal())
            T[i, :3, :3] = (R @ T[i, :3, :3]).astype(np.float32)
            T[i, :3, 3] += rng.normal(size=3).astype(np.float32) * trans_sigma
    prob["pose_Tcw"] = T.reshape(prob["pose_Tcw"].shape).astype(np.float32)
    prob["point_xyz"] = (prob["point_xyz"] + rng.normal(size=prob["point_xyz"].shape) * point_sigma).astype(np.float32)
    return prob


@pytest.mark.parametrize("cfg", [(42, 0.5, 3, 2), (43, 0.5, 3, 2), (43, 1.0, 6, 4), (42, 0.3, 10, 8), (44, 0.5, 3, 2)])
def test_lba_rejected_trials_and_early_termination(pkg, oracle, gpu, cfg):
    """LM trials that fail (rho <= 0): estimates restored from the backup the Schur item kernel made, residuals and
    the speculatively rebuilt system recomputed at the restored estimates, lambda *= nu; ten failures end the
    optimisation (levenberg.cpp:118-150).  Same iteration counts, lambda and outlier sets as the oracle."""
    prob = _hard_problem(pkg, *cfg)
    want = oracle.lba_solve(prob)
    lt, n1 = want["lambda_trace"], want["iters"][0]
    got = pkg.LocalBA().LocalBundleAdjustment(prob)
    assert got["status"] == 0 and got["iters"] == want["iters"]
    assert (got["edge_outlier"] == want["edge_outlier"]).all()
    assert abs(got["final_chi2"] - want["chi2_trace"][-1]) <= 1e-6 * want["chi2_trace"][-1] + 1e-9   # (one case ends at chi2 ~ 1e-27)
    assert close(got["pose_Tcw"], want["pose_Tcw"], 1e-4) and close(got["point_xyz"], want["point_xyz"], 1e-4)
    if cfg[0] != 44:   # these problems do contain rejected trials (lambda grows inside a pass)
        grow = [lt[i + 1] > lt[i] for i in range(len(lt) - 1) if i + 1 != n1]
        assert any(grow)

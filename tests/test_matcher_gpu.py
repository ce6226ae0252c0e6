"""Parity of the HIP ORBmatcher path against the oracle / golden fixtures (indices and counts are
compared exactly)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_hamming_best2_vs_bruteforce(pkg, gpu):
    rng = np.random.default_rng(0)
    S = pkg.synth
    m = pkg.Matcher()
    for nq, nt in ((1, 1), (7, 300), (2000, 2000), (513, 4099)):
        q, t = S.synth_descriptors(rng, nq), S.synth_descriptors(rng, nt)
        if nt > 10:
            t[: min(nq, nt) // 2] = S.flip_bits(rng, q[: min(nq, nt) // 2], 0.05)
            t[5] = t[3]  # exact tie: lowest index must win
        bi, bd, sd = m.hamming_best2(q, t)
        D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(axis=2)
        assert (bi == D.argmin(axis=1)).all() and (bd == D.min(axis=1)).all()
        if nt > 1:
            assert (sd == np.sort(D, axis=1)[:, 1]).all()
        else:
            assert (sd == 256).all()
    z = np.zeros((2, 32), np.uint8)
    o = np.full((3, 32), 255, np.uint8)
    bi, bd, sd = m.hamming_best2(z, o)
    assert (bd == 256).all() and (bi == 0).all()


def test_search_by_bow_vs_oracle(pkg, oracle, gpu):
    S = pkg.synth
    for ratio, ori in ((0.7, True), (0.6, False), (0.9, True)):
        m = pkg.Matcher(ratio, ori)
        probs = [S.synth_bow_problem(s, 900 + 50 * s, 1000 - 30 * s, n_nodes=60 + 10 * s, nnratio=ratio, check_orientation=ori) for s in range(6)]
        res = m.SearchByBoW(probs)
        for p, (n, match) in zip(probs, res):
            on, om = oracle.search_by_bow(p)
            assert n == on and (match == om).all()
    # KITTI-size pair and a big vocabulary bucket (> 64 candidates per node)
    m = pkg.Matcher(0.7, True)
    for p in (S.synth_bow_problem(50, 2000, 2000), S.synth_bow_problem(51, 1500, 1500, n_nodes=6)):
        n, match = m.SearchByBoW(p)
        on, om = oracle.search_by_bow(p)
        assert n == on and (match == om).all()
    g = np.load(os.path.join(GOLD, "bow_300.npz"))
    p = {k: g[k] for k in g.files if k not in ("nmatches", "match")}
    n, match = pkg.Matcher(float(g["nnratio"]), bool(g["check_orientation"])).SearchByBoW(p)
    assert n == int(g["nmatches"]) and (match == g["match"]).all()


def test_search_by_projection_vs_oracle(pkg, oracle, gpu):
    S = pkg.synth
    for seed in range(6):
        f, mp = S.synth_proj_mp_problem(seed, n_f=1000 + 100 * seed, n_mp=1500, th=3.0 if seed % 2 else 1.0)
        on, om = oracle.search_by_projection_mp(f, mp)
        m = pkg.Matcher(float(mp["nnratio"]), True)
        n, match = m.SearchByProjection(f, mp, th=float(mp["th"]))
        assert n == on and (match == om).all()
    f, mp = S.synth_proj_mp_problem(7, n_f=2000, n_mp=3000, w=1241, h=376)
    on, om = oracle.search_by_projection_mp(f, mp)
    n, match = pkg.Matcher(float(mp["nnratio"]), True).SearchByProjection(f, mp, th=float(mp["th"]))
    assert n == on and (match == om).all()


def test_search_windows_from_one_cell_to_the_whole_grid(pkg, oracle, gpu):
    """The window of a query is enumerated one candidate per lane over the runs of the grid's columns: windows of a single
    cell, of all 64 columns with thousands of candidates (many rounds of 64), at the image border, and over a nearly
    empty grid give the reference's candidates in the reference's order (best / second-best ties decide by position)."""
    S = pkg.synth
    for seed, n_f, n_mp, th in ((31, 1200, 120, 150.0), (32, 3000, 60, 400.0), (33, 40, 300, 60.0), (34, 900, 900, 0.2), (35, 2000, 200, 25.0)):
        f, mp = S.synth_proj_mp_problem(seed, n_f=n_f, n_mp=n_mp, th=th)
        on, om = oracle.search_by_projection_mp(f, mp)
        n, match = pkg.Matcher(float(mp["nnratio"]), True).SearchByProjection(f, mp, th=float(mp["th"]))
        assert n == on and (match == om).all(), (seed, n, on)


def test_search_by_projection_entry_pool_from_window_populations(pkg, oracle, gpu, monkeypatch):
    """large local maps size the candidate pool from the real window populations (count pass, scan, fill pass) instead of
    n_mp x n_f slots; forced here on an ordinary problem, also with a budget that is too small (the call learns the total
    from the scan and runs once more, exactly sized): same matches as the one-pass form and the oracle"""
    f, mp = pkg.synth.synth_proj_mp_problem(11, n_f=1500, n_mp=2500, th=4.0)
    on, om = oracle.search_by_projection_mp(f, mp)
    for budget in (None, "300"):
        monkeypatch.setenv("AOS2_PROJ_TWO_PASS", "1")
        if budget:
            monkeypatch.setenv("AOS2_PROJ_POOL_BUDGET", budget)
        n, match = pkg.Matcher(float(mp["nnratio"]), True).SearchByProjection(f, mp, th=float(mp["th"]))
        assert n == on and (match == om).all()
    monkeypatch.delenv("AOS2_PROJ_TWO_PASS")
    monkeypatch.delenv("AOS2_PROJ_POOL_BUDGET")
    n, match = pkg.Matcher(float(mp["nnratio"]), True).SearchByProjection(f, mp, th=float(mp["th"]))
    assert n == on and (match == om).all() and on > 100


def test_search_by_projection_parallel_resolve_equals_sequential(pkg, oracle, gpu, monkeypatch):
    """the fixed-point stage B (all map points decide in parallel from B[f] = first earlier taker, iterated until the
    decisions reproduce themselves) == the one-wave sequential loop (AOS2_SERIAL_RESOLVE=1) == the oracle, also when
    almost every map point competes for the same few features (long dependency chains, many passes)."""
    S = pkg.synth
    cases = [S.synth_proj_mp_problem(40, n_f=1200, n_mp=1500, th=3.0),
             S.synth_proj_mp_problem(41, n_f=60, n_mp=2500, th=12.0),          # ~40 map points per feature
             S.synth_proj_mp_problem(42, n_f=8, n_mp=900, th=40.0, w=160, h=120)]  # every window holds every feature
    for f, mp in cases:
        mp["has_obs"] = (np.arange(len(mp["has_obs"])) % 3 != 0).astype(np.uint8)   # mixes blocking / non-blocking takers
        on, om = oracle.search_by_projection_mp(f, mp)
        n, match = pkg.Matcher(float(mp["nnratio"]), True).SearchByProjection(f, mp, th=float(mp["th"]))
        assert n == on and (match == om).all()
        monkeypatch.setenv("AOS2_SERIAL_RESOLVE", "1")
        n2, match2 = pkg.Matcher(float(mp["nnratio"]), True).SearchByProjection(f, mp, th=float(mp["th"]))
        monkeypatch.delenv("AOS2_SERIAL_RESOLVE")
        assert n2 == on and (match2 == om).all()
    assert on > 0


def test_greedy_searches_parallel_resolve_equals_sequential(pkg, oracle, gpu, monkeypatch):
    """SearchByBoW (KF,F) / (KF,KF), SearchByProjection(Current, Last), SearchByProjection(pKF, Scw) and the
    relocalisation search: fixed-point stage B == the one-wave
    sequential stage B (AOS2_SERIAL_RESOLVE=1) == the oracle, including tiny vocabularies (every keyframe feature
    competes for the same frame features)."""
    S = pkg.synth
    bow = [S.synth_bow_problem(60, 1000, 900, n_nodes=80), S.synth_bow_problem(61, 1200, 300, n_nodes=3),
           S.synth_bow_problem(62, 800, 40, n_nodes=1, nnratio=0.95)]
    bowkf = [S.synth_bow_kf_problem(63, 900, 900, n_nodes=60), S.synth_bow_kf_problem(64, 1000, 200, n_nodes=2, nnratio=0.95)]
    last = [S.synth_proj_last_problem(70, n=1000, th=7.0), S.synth_proj_last_problem(71, n=1200, th=15.0, mono=True)]
    gen = [S.synth_proj_gen_problem(80, n_f=1000, n_pts=1400, cfg="tum", th=10),
           S.synth_proj_gen_problem(81, n_f=120, n_pts=2500, cfg="kitti", th=25)]      # ~20 points per feature
    for serial in (False, True):
        if serial:
            monkeypatch.setenv("AOS2_SERIAL_RESOLVE", "1")
        for p in bow:
            n, match = pkg.Matcher(float(p["nnratio"]), bool(p["check_orientation"])).SearchByBoW(p)
            on, om = oracle.search_by_bow(p)
            assert n == on and (match == om).all()
        for p in bowkf:
            n, match = pkg.Matcher(float(p["nnratio"]), bool(p["check_orientation"])).SearchByBoWKF(p)
            on, om = oracle.search_by_bow_kf(p)
            assert n == on and (match == om).all()
        for cur, p in last:
            n, match = pkg.Matcher(0.9, bool(p["check_orientation"])).SearchByProjectionLast(cur, p, float(p["th"]), int(p["mono"]))
            on, om = oracle.search_by_projection_last(cur, p)
            assert n == on and (match == om).all()
        for f, p in gen:        # SearchByProjection(pKF, Scw, ...) and the relocalisation search
            n, match = pkg.Matcher().SearchByProjectionKF(f, p)
            on, om = oracle.search_by_projection_kf(f, p)
            assert n == on and (match == om).all()
            for ori in (True, False):
                n, match = pkg.Matcher(0.9, ori).SearchByProjectionReloc(f, p, 100)
                on, om = oracle.search_by_projection_reloc(f, p, 100, ori)
                assert n == on and (match == om).all()
    monkeypatch.delenv("AOS2_SERIAL_RESOLVE")


def test_search_by_projection_last_vs_oracle(pkg, oracle, gpu):
    S = pkg.synth
    for seed in range(10):
        cur, p = S.synth_proj_last_problem(seed, n=1000 + 100 * (seed % 3), mono=(seed % 5 == 4), th=7.0 if seed % 2 else 15.0,
                                           check_orientation=(seed != 3))
        on, om = oracle.search_by_projection_last(cur, p)
        m = pkg.Matcher(0.9, bool(p["check_orientation"]))
        n, match = m.SearchByProjectionLast(cur, p, float(p["th"]), int(p["mono"]))
        assert n == on and (match == om).all()
    g = np.load(os.path.join(GOLD, "proj_last_300.npz"))
    f = {k[2:]: g[k] for k in g.files if k.startswith("f_")}
    pl = {k[3:]: g[k] for k in g.files if k.startswith("pl_")}
    f["n_f"], f["n_levels"] = int(f["n_f"]), int(f["n_levels"])
    n, match = pkg.Matcher(0.9, bool(pl["check_orientation"])).SearchByProjectionLast(f, pl, float(pl["th"]), int(pl["mono"]))
    assert n == int(g["nmatches"]) and (match == g["match"]).all()


def test_match_on_extracted_frames(pkg, oracle, gpu):
    """extract -> brute-force match of a frame against a shifted copy: most keypoints re-match."""
    img = pkg.synth.synth_image(77)
    ex = pkg.Extractor()
    k0, d0 = ex(img)
    k1, d1 = ex(np.roll(img, 3, axis=1))
    bi, bd, sd = pkg.Matcher().hamming_best2(d0, d1)
    good = bd <= 50
    assert good.mean() > 0.5
    dx = k1["x"][bi[good]] - k0["x"][good]
    assert np.median(np.abs(dx - 3 * 1.0)) < 4.0


def test_search_by_bow_kf_vs_oracle(pkg, oracle, gpu):
    """SearchByBoW(KF, KF) src/ORBmatcher.cc:522-655"""
    S = pkg.synth
    for ratio, ori in ((0.75, True), (0.6, False), (0.9, True)):
        m = pkg.Matcher(ratio, ori)
        probs = [S.synth_bow_kf_problem(s, 900 + 40 * s, 1000 - 25 * s, n_nodes=50 + 10 * s, nnratio=ratio, check_orientation=ori)
                 for s in range(6)]
        for p, (n, match) in zip(probs, m.SearchByBoWKF(probs)):
            on, om = oracle.search_by_bow_kf(p)
            assert n == on and (match == om).all() and n > 100
    # no map points on one side / empty keyframes
    m = pkg.Matcher(0.75, True)
    p = S.synth_bow_kf_problem(9, 300, 300)
    p["has_mp2"] = np.zeros_like(p["has_mp2"])
    n, match = m.SearchByBoWKF(p)
    assert n == 0 and (match == -1).all() and oracle.search_by_bow_kf(p)[0] == 0
    g = np.load(os.path.join(GOLD, "bow_kf_300.npz"))
    p = {k: g[k] for k in g.files if k not in ("nmatches", "match")}
    n, match = pkg.Matcher(float(g["nnratio"]), bool(g["check_orientation"])).SearchByBoWKF(p)
    assert n == int(g["nmatches"]) and (match == g["match"]).all()


def test_search_for_triangulation_vs_oracle(pkg, oracle, gpu):
    """SearchForTriangulation src/ORBmatcher.cc:657-823: stereo, stereo-only and monocular keyframes"""
    S = pkg.synth
    for ori in (True, False):
        m = pkg.Matcher(0.6, ori)
        for kw in (dict(), dict(only_stereo=True), dict(mono=True), dict(cfg="tum")):
            probs = [S.synth_triang_problem(s, 1000 + 50 * s, 1100 - 40 * s, n_nodes=40 + 15 * s, check_orientation=ori, **kw)
                     for s in range(5)]
            res = m.SearchForTriangulation(probs, only_stereo=kw.get("only_stereo", False))
            for p, (n, match) in zip(probs, res):
                on, om = oracle.search_for_triangulation(p)
                assert n == on and (match == om).all()
            assert sum(r[0] for r in res) > (50 if kw.get("only_stereo") else 300)
    # equal distances: the LAST candidate of the bucket wins (dist > bestDist is the skip test, :734)
    p = S.synth_triang_problem(77, 200, 200, n_nodes=1)
    p["has_mp1"][:] = 0
    p["has_mp2"][:] = 0
    p["desc2"][:] = p["desc1"][0]
    p["F12"] = np.zeros(9, np.float32)
    p["F12"][6] = 1e-3  # a = 1e-3, b = 0: every point lies near the "line"
    p["u_right1"][:] = 1.0
    m = pkg.Matcher(0.6, False)
    n, match = m.SearchForTriangulation(p)
    on, om = oracle.search_for_triangulation(p)
    assert n == on and (match == om).all()
    g = np.load(os.path.join(GOLD, "triang_300.npz"))
    p = {k: g[k] for k in g.files if k not in ("nmatches", "match")}
    n, match = pkg.Matcher(0.6, bool(g["check_orientation"])).SearchForTriangulation(p, only_stereo=bool(g["only_stereo"]))
    assert n == int(g["nmatches"]) and (match == g["match"]).all()


def test_compute_distinctive_descriptors_vs_oracle(pkg, oracle, gpu):
    """MapPoint::ComputeDistinctiveDescriptors src/MapPoint.cc:275-340 (batched)"""
    S = pkg.synth
    m = pkg.Matcher()
    for seed, n, mo in ((0, 3000, 24), (1, 500, 70), (2, 40, 3)):
        off, desc = S.synth_observations(seed, n, mo)
        best = m.ComputeDistinctiveDescriptors(off, desc)
        assert (best == oracle.compute_distinctive_descriptors(off, desc)).all()
        cnt = np.diff(off)
        assert ((best == -1) == (cnt == 0)).all() and (best[cnt > 0] < cnt[cnt > 0]).all()
    # ties: identical descriptors -> index 0; two descriptors -> median = row[0] = 0 for both -> index 0
    off = np.array([0, 5, 7, 8, 8], np.int32)
    desc = np.zeros((8, 32), np.uint8)
    desc[5] = 255
    assert m.ComputeDistinctiveDescriptors(off, desc).tolist() == [0, 0, 0, -1]
    assert len(m.ComputeDistinctiveDescriptors(np.zeros(1, np.int32), np.zeros((0, 32), np.uint8))) == 0


def test_fuse_vs_oracle(pkg, oracle, gpu):
    """search part of Fuse(pKF, vpMapPoints, th) :825-975 and Fuse(pKF, Scw, ...) :977-1100"""
    S = pkg.synth
    m = pkg.Matcher()
    tot = 0
    for seed, cfg, th, stereo in ((0, "kitti", 3.0, True), (1, "tum", 3.0, True), (2, "euroc", 4.0, False), (3, "kitti", 2.5, True)):
        f, p = S.synth_proj_gen_problem(seed, n_f=1000 + 200 * seed, n_pts=1500 + 300 * seed, cfg=cfg, th=th, stereo=stereo)
        for sim3 in (False, True):
            n, bi, bd = m.Fuse(f, p, sim3=sim3)
            on, obi, obd = oracle.fuse(f, p, sim3=sim3)
            assert n == on and (bi == obi).all() and (bd == obd).all()
            tot += n
            # every fused pair respects the gates: TH_LOW, level window
            k = np.flatnonzero(bi >= 0)
            assert (bd[k] <= 50).all() and p["valid"][k].all()
    assert tot > 1000
    f, p = S.synth_proj_gen_problem(9, n_f=300, n_pts=0)
    assert m.Fuse(f, p)[0] == 0
    g = np.load(os.path.join(GOLD, "fuse_400.npz"))
    f = {k[2:]: g[k] for k in g.files if k.startswith("f_")}
    p = {k[2:]: g[k] for k in g.files if k.startswith("p_")}
    f["n_f"], f["n_levels"], p["n_pts"] = int(f["n_f"]), int(f["n_levels"]), int(p["n_pts"])
    n, bi, bd = m.Fuse(f, p)
    assert n == int(g["n"]) and (bi == g["best_idx"]).all() and (bd == g["best_dist"]).all()


def test_search_by_projection_kf_and_reloc_vs_oracle(pkg, oracle, gpu):
    """SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) :290-403 and the relocalisation search :1472-1599"""
    S = pkg.synth
    for seed in range(5):
        f, p = S.synth_proj_gen_problem(20 + seed, n_f=900 + 150 * seed, n_pts=1400, cfg=("kitti", "tum")[seed % 2], th=[10, 4, 7][seed % 3])
        m = pkg.Matcher()
        n, match = m.SearchByProjectionKF(f, p)
        on, om = oracle.search_by_projection_kf(f, p)
        assert n == on and (match == om).all() and n > 200
        assert not f["f_mp_state"][match >= 0].any()       # slots that were taken on entry stay untouched
        for ori in (True, False):
            for orb_dist in (100, 64):
                mm = pkg.Matcher(0.9, ori)
                n, match = mm.SearchByProjectionReloc(f, p, orb_dist)
                on, om = oracle.search_by_projection_reloc(f, p, orb_dist, ori)
                assert n == on and (match == om).all()
                assert n == (match >= 0).sum() and (ori or (match != -2).all())
    g = np.load(os.path.join(GOLD, "reloc_400.npz"))
    f = {k[2:]: g[k] for k in g.files if k.startswith("f_")}
    p = {k[2:]: g[k] for k in g.files if k.startswith("p_")}
    f["n_f"], f["n_levels"], p["n_pts"] = int(f["n_f"]), int(f["n_levels"]), int(p["n_pts"])
    n, match = pkg.Matcher(0.9, True).SearchByProjectionReloc(f, p, 100)
    assert n == int(g["n"]) and (match == g["match"]).all()


def test_search_by_sim3_vs_oracle(pkg, oracle, gpu):
    """SearchBySim3 :1102-1326"""
    S = pkg.synth
    m = pkg.Matcher()
    tot = 0
    for seed in range(5):
        f1, f2, p12, p21 = S.synth_sim3_problem(seed, 900 + 100 * seed, 1000 - 50 * seed, cfg=("kitti", "euroc")[seed % 2])
        n, match = m.SearchBySim3(f1, f2, p12, p21)
        on, om = oracle.search_by_sim3(f1, f2, p12, p21)
        assert n == on and (match == om).all() and n == (match >= 0).sum()
        tot += n
        k = np.flatnonzero(match >= 0)
        assert len(set(match[k].tolist())) == len(k)       # mutual agreement makes the assignment injective
    assert tot > 500


def test_search_for_initialization_vs_oracle(pkg, oracle, gpu):
    """SearchForInitialization :405-520 (vMatchedDistance gate, match stealing, rotation check)"""
    S = pkg.synth
    tot = 0
    for seed in range(5):
        f2, q = S.synth_init_problem(seed, 1200 + 200 * seed, 1500 - 100 * seed)
        for ratio, ori, ws in ((0.9, True, 100), (0.7, False, 30), (0.9, True, 10)):
            n, match = pkg.Matcher(ratio, ori).SearchForInitialization(f2, q, ws)
            on, om = oracle.search_for_initialization(f2, q, ws, ratio, ori)
            assert n == on and (match == om).all() and n == (match >= 0).sum()
            k = np.flatnonzero(match >= 0)
            assert len(set(match[k].tolist())) == len(k) and (q["octave1"][k] == 0).all() and (f2["kp_octave"][match[k]] == 0).all()
            tot += n
    assert tot > 1500
    f2, q = S.synth_init_problem(9, 50, 60)
    q0 = {k: v[:0] for k, v in q.items()}
    assert pkg.Matcher().SearchForInitialization(f2, q0)[0] == 0
    g = np.load(os.path.join(GOLD, "init_500.npz"))
    f2 = {k[2:]: g[k] for k in g.files if k.startswith("f_")}
    q = {k[2:]: g[k] for k in g.files if k.startswith("q_")}
    f2["n_f"], f2["n_levels"] = int(f2["n_f"]), int(f2["n_levels"])
    n, match = pkg.Matcher(0.9, True).SearchForInitialization(f2, q, 100)
    assert n == int(g["n"]) and (match == g["match"]).all()


def test_is_in_frustum_feeds_search_local_points(pkg, oracle, gpu):
    """Frame::isInFrustum (src/Frame.cc:298-354) on the device, and its outputs driving SearchByProjection(F, vpMP):
    the Tracking::SearchLocalPoints chain equals the oracle chain."""
    S = pkg.synth
    m = pkg.Matcher(0.8, True)
    seen = 0
    for seed, cfg in ((0, "kitti"), (1, "tum"), (2, "euroc")):
        f, p = S.synth_proj_gen_problem(60 + seed, n_f=1200, n_pts=2500, cfg=cfg)
        for lim in (0.5, 0.9):
            a, b = m.isInFrustum(f, p, lim), oracle.is_in_frustum(f, p, lim)
            for k in a:
                assert a[k].tobytes() == b[k].tobytes(), k
            seen += int(a["track_in_view"].sum())
        a = m.isInFrustum(f, p, 0.5)
        # gates visible in the outputs: closed image bounds, cosine limit, level range
        iv = a["track_in_view"] > 0
        assert (a["view_cos"][iv] >= 0.5).all() and (a["pred_level"][iv] >= 0).all() and (a["pred_level"][iv] < f["n_levels"]).all()
        assert (a["proj_x"][iv] >= f["min_x"]).all() and (a["proj_x"][iv] <= f["max_x"]).all()
        mp = dict(n_mp=p["n_pts"], track_in_view=a["track_in_view"], pred_level=a["pred_level"], view_cos=a["view_cos"],
                  proj_x=a["proj_x"], proj_y=a["proj_y"], proj_xr=a["proj_xr"], desc=p["desc"],
                  has_obs=np.ones(p["n_pts"], np.uint8), th=np.float32(3.0), nnratio=np.float32(0.8))
        n, match = m.SearchByProjection(f, mp, th=3.0)
        on, om = oracle.search_by_projection_mp(f, mp)
        assert n == on and (match == om).all() and n > 100
    assert seen > 3000
    empty = {k: (v[:0] if isinstance(v, np.ndarray) and v.ndim >= 1 and len(v) == p["n_pts"] else v) for k, v in p.items()}
    empty["n_pts"] = 0
    assert len(m.isInFrustum(f, empty)["proj_x"]) == 0


def test_is_in_frustum_hand_computed_levels(pkg, gpu):
    """range gate on 1.2f * mfMaxDistance / 0.8f * mfMinDistance, PredictScale on the raw mfMaxDistance (src/Frame.cc:326-343,
    src/MapPoint.cc:413-459): hand-computed levels, on the single-frame entry and through the Fuse / KF / reloc gates"""
    f, p, want_in, want_lvl = pkg.synth.frustum_hand_case()
    a = pkg.Matcher(0.8, True).isInFrustum(f, p, 0.5)
    assert (a["track_in_view"] == want_in).all()
    assert (a["pred_level"][want_in > 0] == want_lvl[want_in > 0]).all()


def test_frame_grid_and_rgbd_depth_vs_oracle(pkg, oracle, gpu):
    """Frame::AssignFeaturesToGrid (src/Frame.cc:259-274) and Frame::ComputeStereoFromRGBD (:672-693)"""
    S = pkg.synth
    m = pkg.Matcher()
    rng = np.random.default_rng(4)
    for n, (w, h) in ((1000, (640, 480)), (2000, (1241, 376)), (1, (640, 480)), (0, (640, 480)), (5000, (752, 480))):
        x = rng.uniform(-3, w + 3, n).astype(np.float32)      # undistorted keypoints can leave the image (:417)
        y = rng.uniform(-3, h + 3, n).astype(np.float32)
        if n > 10:
            x[:5], y[:5] = x[5], y[5]                        # several features in one cell keep their index order
        gwi, ghi = np.float32(64.0 / w), np.float32(48.0 / h)
        off, idx = m.AssignFeaturesToGrid(x, y, 0.0, 0.0, gwi, ghi)
        ooff, oidx = oracle.assign_features_to_grid(x, y, 0.0, 0.0, gwi, ghi)
        assert (off == ooff).all() and (idx == oidx).all() and off[-1] == len(idx) <= n
        # identical to the python builder the other tests use, on in-image points
        if n >= 1000:
            xi, yi = np.clip(x, 1, w - 1), np.clip(y, 1, h - 1)
            o2, i2, _, _ = S.build_grid(xi, yi, np.float32(0), np.float32(0), np.float32(w), np.float32(h))
            o3, i3 = m.AssignFeaturesToGrid(xi, yi, 0.0, 0.0, gwi, ghi)
            assert (o2 == o3).all() and (i2 == i3).all()
    depth = rng.uniform(0.3, 8.0, (480, 640)).astype(np.float32)
    depth[rng.random(depth.shape) < 0.2] = 0.0               # missing depth
    kx = rng.uniform(0, 639.9, 1500).astype(np.float32)
    ky = rng.uniform(0, 479.9, 1500).astype(np.float32)
    kux = (kx + rng.normal(0, 0.3, 1500)).astype(np.float32)
    ur, dp = m.ComputeStereoFromRGBD(kx, ky, kux, depth, 40.0)
    our, odp = oracle.stereo_from_rgbd(kx, ky, kux, depth, 40.0)
    assert ur.tobytes() == our.tobytes() and dp.tobytes() == odp.tobytes()
    assert ((dp > 0) == (depth[ky.astype(int), kx.astype(int)] > 0)).all() and (ur[dp < 0] == -1).all()
    with pytest.raises(pkg.AosError):
        m.ComputeStereoFromRGBD(np.array([700.0], np.float32), np.array([10.0], np.float32), np.array([700.0], np.float32), depth, 40.0)


def test_search_by_projection_batch_equals_single(pkg, oracle, gpu):
    """n frames in one launch give exactly the per-frame results (and the oracle's)"""
    S = pkg.synth
    probs = [S.synth_proj_mp_problem(200 + s, n_f=900 + 37 * s, n_mp=1200 + 50 * s, th=3.0) for s in range(12)]
    frames, mps = [p[0] for p in probs], [p[1] for p in probs]
    m = pkg.Matcher(float(mps[0]["nnratio"]), True)
    res = m.SearchByProjectionBatch(frames, mps, th=3.0)
    ms_batch = m.last_device_ms()
    for (f, mp), (n, match) in zip(probs, res):
        on, om = oracle.search_by_projection_mp(f, mp)
        assert n == on and (match == om).all()
    n1, m1 = m.SearchByProjection(frames[0], mps[0], th=3.0)
    assert n1 == res[0][0] and (m1 == res[0][1]).all()
    assert ms_batch < 12 * m.last_device_ms()       # 12 frames in one call cost less than 12 single calls


def test_one_handle_across_growing_and_shrinking_calls(pkg, oracle, gpu):
    """The handle's page-locked input buffer and device arena persist between calls and grow on demand -- also in the
    middle of a call (a 40-pair SearchByBoW pushes ~6 MB into a 1 MB buffer).  Small, large, small again, different entry
    points interleaved: every result equals the oracle's."""
    S = pkg.synth
    f, mp = S.synth_proj_mp_problem(7, n_f=1000, n_mp=1500)
    m = pkg.Matcher(float(mp["nnratio"]), True)
    small = S.synth_bow_problem(70, 200, 220, n_nodes=20, nnratio=float(mp["nnratio"]), check_orientation=True)
    big = [S.synth_bow_problem(100 + s, 1000, 1000, nnratio=float(mp["nnratio"]), check_orientation=True) for s in range(40)]
    for round_ in range(2):
        n, match = m.SearchByBoW(small)
        on, om = oracle.search_by_bow(small)
        assert n == on and (match == om).all()
        res = m.SearchByBoW(big)
        for p, (n, match) in list(zip(big, res))[::7]:
            on, om = oracle.search_by_bow(p)
            assert n == on and (match == om).all()
        n, match = m.SearchByProjection(f, mp, th=float(mp["th"]))
        on, om = oracle.search_by_projection_mp(f, mp)
        assert n == on and (match == om).all()
        n, match = m.SearchByBoW(small)
        on, om = oracle.search_by_bow(small)
        assert n == on and (match == om).all()


def test_projection_searches_beyond_a_gibibyte_of_candidates(pkg, oracle, gpu):
    """SearchByProjection(pKF, Scw, ...) and the relocalisation search give every point a slice of n_f candidate entries: 20 000
    loop-closure points x 8000 features = 1.3 GB of entries -- rejected with an error until round 3 (a 1 GiB cap written for a
    smaller device), now limited only by the 32-bit entry offsets (16 GiB) and the allocation: same results as the oracle"""
    f, p = pkg.synth.synth_proj_gen_problem(91, n_f=8000, n_pts=20000, cfg="kitti", th=10.0)
    m = pkg.Matcher(0.75, True)
    n, match = m.SearchByProjectionKF(f, p)
    on, om = oracle.search_by_projection_kf(f, p)
    assert n == on and (match == om).all() and n > 500
    n, match = m.SearchByProjectionReloc(f, p, 100)
    on, om = oracle.search_by_projection_reloc(f, p, 100, True)
    assert n == on and (match == om).all() and n > 500

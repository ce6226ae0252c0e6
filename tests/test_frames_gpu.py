"""Device-resident frame batches (aos2_frames_*): every stage of the per-frame tracking chain against the oracle's
host chain (oracle/chain.py) on the same scenario: integer results (map point assignments, outlier flags, grids, match
counts) bit-identical, mvuRight / mvDepth bit-identical, poses within 1e-5 (float32 write-back of an f64 solve)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5


def close(a, b, key=None):
    import parity   # oracle/parity.py: |a - b| <= 1e-5, literally; records the worst difference seen
    ok = parity.close(a, b, TOL, key=key)
    if not ok:
        print("worst |difference|:", parity.worst(a, b))
    return ok


@pytest.fixture(scope="module")
def ochain(oracle):
    sys.path.insert(0, os.path.dirname(oracle.__file__))
    import chain
    return chain


TUM1_DIST = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]   # Examples/RGB-D/TUM1.yaml: Camera.k1 k2 p1 p2 k3


# euroc: 752 x 480, 1200 features (> 1024 slots per frame); the last two: frames with lens distortion (mvKeysUn != mvKeys,
# image bounds = undistorted corners; the synthetic images are not warped, so fewer map points land near their features)
@pytest.mark.parametrize("seed,batch,cfg,dist", [(1, 6, "tum", None), (2, 3, "tum", None), (3, 2, "euroc", None),
                                                 (4, 3, "tum", [0.02, -0.01, 0.001, -0.0005, 0.0]), (5, 2, "tum", TUM1_DIST)])
def test_chain_stage_by_stage_vs_oracle(pkg, oracle, ochain, gpu, seed, batch, cfg, dist):
    import torch
    scen = pkg.scenario.tracking_scenario(seed, batch, cfg=cfg, n_unique=batch, dist=dist)
    tc = pkg.chain.TrackingChain(scen, n_local=1500)
    B, W, H, cap = tc.B, tc.W, tc.H, tc.cap
    F = pkg.capi.Frames
    # --- extraction + Frame::Frame
    tc.ex.extract_batch_device(tc.d_cur.data_ptr(), B, W, H, W, W * H, tc.d_kps.data_ptr(), tc.d_desc.data_ptr(), cap, tc.d_n.data_ptr())
    c = tc.cur
    s = scen
    c.build(tc.ex, tc.d_kps.data_ptr(), tc.d_desc.data_ptr(), tc.d_n.data_ptr(), W, H, tc.d_depth.data_ptr(),
            float(s["fx"]), float(s["fy"]), float(s["cx"]), float(s["cy"]), float(s["mbf"]))
    c.set_pose(tc.d_guess.data_ptr())
    n = tc.d_n.cpu().numpy()
    kps = tc.d_kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28).copy().view(pkg.capi.KP_DTYPE).reshape(B, cap)
    desc = tc.d_desc.cpu().numpy()
    oe = oracle.Extractor(nfeatures=scen["nfeatures"])
    sf, isg = oe.scale_factors, oe.inv_sigma2
    frames = []
    ur, dp, goff, gidx = c.get(F.U_RIGHT), c.get(F.DEPTH), c.get(F.GRID_OFF), c.get(F.GRID_IDX)
    kun_x, kun_y = c.get(F.KEYS_UN_X), c.get(F.KEYS_UN_Y)
    for b in range(B):
        okps, odesc = oe.extract(scen["cur"][b])
        assert len(okps) == n[b] and okps.tobytes() == kps[b, :n[b]].tobytes() and (odesc == desc[b, :n[b]]).all()
        f = ochain.frame_from_extraction(okps, odesc, scen["depth_cur"][b], scen, sf, isg)
        frames.append(f)
        assert ur[b, :n[b]].tobytes() == f["u_right"].tobytes() and dp[b, :n[b]].tobytes() == f["depth"].tobytes()
        assert (goff[b] == f["grid_off"]).all() and (gidx[b, :goff[b, -1]] == f["grid_idx"][:goff[b, -1]]).all()
        assert (f["u_right"] < 0).any() and (f["u_right"] >= 0).any()   # mono and stereo observations both occur
        assert kun_x[b, :n[b]].tobytes() == f["kp_x"].tobytes() and kun_y[b, :n[b]].tobytes() == f["kp_y"].tobytes()   # mvKeysUn
        if dist is not None:
            assert (f["kp_x"] != okps["x"]).any()
            bd = pkg.capi.frame_image_bounds(W, H, s["fx"], s["fy"], s["cx"], s["cy"], dist)
            assert (bd == np.array([f["min_x"], f["max_x"], f["min_y"], f["max_y"]], np.float32)).all()
    # LastFrame members of the oracle chain
    last_h = []
    for b in range(B):
        lk, _ = tc.host_last[b]
        nl = len(lk)
        last_h.append(dict(mp=tc.last_mp[b, :nl], outlier=tc.last_outlier[b, :nl], kp_octave=lk["octave"], kp_angle=lk["angle"]))
    table = tc.map["table"]
    want = [ochain.track_frame(frames[b], last_h[b], table, tc.map["local"][b], scen["Tcw_guess"][b], scen["Tlw"][b], scen) for b in range(B)]
    # --- SearchByProjection(CurrentFrame, LastFrame)
    c.SearchByProjectionLast(tc.last, tc.table, tc.th_last, mono=False, check_orientation=True, d_nmatches=tc.d_nm[0].data_ptr())
    mp = c.get(F.MAP_POINTS)
    nm = tc.d_nm.cpu().numpy()
    for b in range(B):
        assert nm[0, b] == want[b]["nmatches_last"] and (nm[0, b] > 100 or dist is not None)
        assert (mp[b, :n[b]] == want[b]["mp_after_last"]).all() and (mp[b, n[b]:] == -1).all()
    # --- PoseOptimization + discard
    c.PoseOptimization(tc.table, tc.d_nm[1].data_ptr())
    T1, o1 = c.get(F.TCW), c.get(F.OUTLIER)
    nm = tc.d_nm.cpu().numpy()
    for b in range(B):
        assert nm[1, b] == want[b]["inliers_1"]
        assert (o1[b, :n[b]] == want[b]["outlier_1"]).all()
        assert close(T1[b], want[b]["Tcw_1"])
        # the optimised pose is close to the true one (the scenario is consistent)
        assert dist is not None or np.abs(T1[b].reshape(4, 4)[:3, 3] - scen["Tcw_true"][b][:3, 3]).max() < 0.02
    c.discard_outliers()
    mp = c.get(F.MAP_POINTS)
    for b in range(B):
        assert (mp[b, :n[b]] == want[b]["mp_after_discard"]).all()
    # --- SearchLocalPoints
    c.SearchLocalPoints(tc.table, tc.d_local.data_ptr(), tc.n_local, tc.th_local, tc.nnratio_local, tc.d_nm[2].data_ptr())
    mp = c.get(F.MAP_POINTS)
    nm = tc.d_nm.cpu().numpy()
    for b in range(B):
        assert nm[2, b] == want[b]["nmatches_local"] and (nm[2, b] > 20 or dist is not None)
        assert (mp[b, :n[b]] == want[b]["mp_after_local"]).all()
    # --- PoseOptimization
    c.PoseOptimization(tc.table, tc.d_nm[3].data_ptr())
    T2, o2 = c.get(F.TCW), c.get(F.OUTLIER)
    nm = tc.d_nm.cpu().numpy()
    for b in range(B):
        assert nm[3, b] == want[b]["inliers_2"]
        assert (o2[b, :n[b]] == want[b]["outlier_2"]).all()
        assert close(T2[b], want[b]["Tcw_2"])
    c.wait()
    # --- the same chain enqueued in one go (asynchronously behind the extractor) gives the same members
    tc.step()
    tc.wait()
    assert (c.get(F.MAP_POINTS) == mp).all() and (c.get(F.OUTLIER) == o2).all() and c.get(F.TCW).tobytes() == T2.tobytes()
    assert (tc.d_nm.cpu().numpy() == nm).all()


def test_chain_tiled_batch_and_window_budget(pkg, gpu, monkeypatch):
    """a tiled batch (frames repeat) gives repeated results; an entry pool that is too small is reported, not ignored"""
    scen = pkg.scenario.tracking_scenario(3, 16, n_unique=4)
    tc = pkg.chain.TrackingChain(scen, n_local=1200)
    tc.step()
    tc.wait()
    F = pkg.capi.Frames
    mp, T = tc.cur.get(F.MAP_POINTS), tc.cur.get(F.TCW)
    for b in range(4, 16):
        assert (mp[b] == mp[b % 4]).all() and T[b].tobytes() == T[b % 4].tobytes()
    assert (tc.d_nm.cpu().numpy()[3] > 50).all()
    monkeypatch.setenv("AOS2_FRAMES_WINDOW_BUDGET", "1")
    tc2 = pkg.chain.TrackingChain(scen, n_local=1200)
    tc2.step()
    with pytest.raises(pkg.AosError):
        tc2.wait()


def test_chain_with_blank_frames(pkg, gpu):
    """a blank / constant current frame has no keypoints: no matches, PoseOptimization returns 0 before touching the pose
    (src/Optimizer.cc:355-356), the other frames of the batch are unaffected"""
    scen = pkg.scenario.tracking_scenario(9, 3, n_unique=3)
    ref = pkg.chain.TrackingChain(scen, n_local=800)
    ref.step()
    ref.wait()
    F = pkg.capi.Frames
    T_ref, mp_ref, nm_ref = ref.cur.get(F.TCW), ref.cur.get(F.MAP_POINTS), ref.d_nm.cpu().numpy()
    scen["cur"][1][:] = 0
    scen["cur"][2][:, :] = scen["cur"][2][0, 0]
    tc = pkg.chain.TrackingChain(scen, n_local=800)
    tc.step()
    tc.wait()
    n, nm, T, mp = tc.d_n.cpu().numpy(), tc.d_nm.cpu().numpy(), tc.cur.get(F.TCW), tc.cur.get(F.MAP_POINTS)
    assert n[0] > 500 and n[1] == 0 and n[2] == 0
    assert (nm[:, 1:] == 0).all() and (nm[:, 0] == nm_ref[:, 0]).all() and nm[3, 0] > 100
    for b in (1, 2):
        assert np.array_equal(T[b], scen["Tcw_guess"][b].reshape(16)) and (mp[b] == -1).all()
    assert T[0].tobytes() == T_ref[0].tobytes() and (mp[0] == mp_ref[0]).all()


def test_rolling_last_frame_needs_no_host_wait(pkg, gpu):
    """In a tracking loop the previous CurrentFrame batch becomes LastFrame while its PoseOptimization / SearchLocalPoints are
    still in flight on its own stream: SearchByProjection(Current, Last) orders itself behind them on the device (no host wait
    in between) and gives what the fully synchronous sequence gives."""
    import torch
    scen = pkg.scenario.tracking_scenario(11, 48, n_unique=8)
    tc = pkg.chain.TrackingChain(scen, n_local=1200)
    B, W, H, cap, s = tc.B, tc.W, tc.H, tc.cap, scen
    F = pkg.capi.Frames
    nxt = F(B, cap)
    ex2 = pkg.Extractor(nfeatures=scen["nfeatures"])
    k2 = torch.zeros((B, cap, 7), dtype=torch.float32, device=tc.dev)
    d2 = torch.zeros((B, cap, 32), dtype=torch.uint8, device=tc.dev)
    n2 = torch.zeros((B,), dtype=torch.int32, device=tc.dev)
    nm = torch.zeros((2, B), dtype=torch.int32, device=tc.dev)

    def run(sync):
        tc.step()
        if sync:
            tc.wait()
        # the camera stands still: the next frames are the same images, their pose guess is the (not yet finished) pose of
        # the frames that now play LastFrame -- read from `tc.cur` on the device by the search itself
        ex2.extract_batch_device_async(tc.d_cur.data_ptr(), B, W, H, W, W * H, k2.data_ptr(), d2.data_ptr(), cap, n2.data_ptr())
        nxt.build(ex2, k2.data_ptr(), d2.data_ptr(), n2.data_ptr(), W, H, tc.d_depth.data_ptr(), float(s["fx"]), float(s["fy"]),
                  float(s["cx"]), float(s["cy"]), float(s["mbf"]))
        nxt.set_pose(tc.d_guess.data_ptr())
        nxt.SearchByProjectionLast(tc.cur, tc.table, 15.0, mono=False, check_orientation=True, d_nmatches=nm[0].data_ptr())
        nxt.PoseOptimization(tc.table, nm[1].data_ptr())
        nxt.wait()
        tc.wait()
        return nxt.get(F.MAP_POINTS), nxt.get(F.TCW), nxt.get(F.OUTLIER), nm.cpu().numpy().copy()

    want = run(True)
    assert (want[3][0] > 100).all()
    for _ in range(3):
        got = run(False)
        for a, b in zip(got, want):
            assert a.tobytes() == b.tobytes()
    # a caller-produced input on a stream of its own: the batch is ordered behind it on the device
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g2 = tc.d_guess.clone()
    nxt.wait_for_stream(side.cuda_stream)
    nxt.set_pose(g2.data_ptr())
    nxt.wait()
    assert nxt.get(F.TCW).tobytes() == tc.d_guess.cpu().numpy().tobytes()


def test_next_image_extracted_beside_the_tracking_of_this_one(pkg, gpu):
    """ONE sequence, frame after frame (chain.step_pipelined): the next image's ORBextractor::operator() is enqueued while the current
    frame is tracked, into the other output buffer set -- every frame's results are those of the plain enqueue-and-wait chain, bit
    for bit."""
    scen = pkg.scenario.tracking_scenario(5, 1, n_unique=1)
    tc = pkg.chain.TrackingChain(scen, n_local=1500)
    F = pkg.capi.Frames
    tc.step()
    tc.wait()
    want = (tc.cur.get(F.MAP_POINTS), tc.cur.get(F.TCW), tc.cur.get(F.OUTLIER), tc.d_nm.cpu().numpy().copy())
    assert want[3][1, 0] > 100 and want[3][3, 0] > 100
    for _ in range(6):
        tc.step_pipelined()
        tc.wait_frame()
        got = (tc.cur.get(F.MAP_POINTS), tc.cur.get(F.TCW), tc.cur.get(F.OUTLIER), tc.d_nm.cpu().numpy().copy())
        for a, b in zip(got, want):
            assert a.tobytes() == b.tobytes()
    tc.wait()


def test_bench_scenario_all_frames_vs_oracle(pkg, oracle, gpu):
    """Exactly what bench.py times: `tracking_scenario(100, 256, n_unique=32)` through TrackingChain.step() (the whole chain
    enqueued at once, asynchronously behind the extractor, n_local = 1500) -- every one of the 256 batch positions (32
    distinct pairs, tiled) against the oracle chain: keypoints, descriptors, mvuRight / mvDepth, the match and inlier counts
    of all four stages, mvpMapPoints, mvbOutlier bit-identical, mTcw within 1e-5.  Twice, with two steps in flight on two
    pipelines like the bench."""
    sys.path.insert(0, os.path.dirname(oracle.__file__))
    import parity
    scen = pkg.scenario.tracking_scenario(100, 256, cfg="tum", n_unique=32)
    pipes = [pkg.chain.TrackingChain(scen, n_local=1500) for _ in range(2)]
    for p in pipes:
        p.ex.set_chunks(1)
    co = parity.ChainOracle(scen, pipes[0])
    for rnd in range(2):
        for p in pipes:
            p.step()
        for p in pipes:
            p.wait()
        for j, p in enumerate(pipes):
            bad = parity.chain_mismatches(parity.chain_snapshot(pkg, p), co, range(256))
            assert bad == [], (rnd, j, bad[:5])
    nm = pipes[0].d_nm.cpu().numpy()
    assert (nm[0] > 100).all() and (nm[3] > 100).all()


def test_reference_keyframe_bow_leg_vs_oracle(pkg, oracle, gpu):
    """Tracking::TrackReferenceKeyFrame's front part on device-resident frames (src/Tracking.cc:858-866): Frame::ComputeBoW =
    the vocabulary transform on the extractor's device output, ordered behind the extraction on the device, then
    SearchByBoW(reference keyframe, frame) with descriptors and keys used in place (aos2_matcher_search_by_bow_device) --
    match arrays and counts equal the oracle's transform + SearchByBoW on its own extraction; run twice behind two steps."""
    sys.path.insert(0, os.path.dirname(oracle.__file__))
    import parity
    scen = pkg.scenario.tracking_scenario(21, 12, n_unique=6)
    tc = pkg.chain.TrackingChain(scen, n_local=1000)
    voc = pkg.synth.synth_vocabulary(401, 10, 4)
    leg = pkg.chain.ReferenceKeyFrameBoW(tc, voc, 10)
    co = parity.ChainOracle(scen, tc)
    for _ in range(2):
        tc.step()
        leg.order()
        leg.run()
        tc.wait()
        res = leg.get_results()
        assert parity.bow_leg_mismatches(res, co, voc, range(10)) == []
        assert all(n > 40 for n, _ in res), [n for n, _ in res]
    # the device-input form equals the host-pointer form on the same arrays
    b = 3
    kd, kk = tc.host_last[b][1], tc.host_last[b][0]
    n_f = int(tc.d_n[b].item())
    fd = tc.d_desc[b, :n_f].cpu().numpy()
    fa = tc.d_kps[b, :n_f, 3].cpu().numpy()
    kb, fb = pkg.Vocabulary(), None
    kb.set_nodes(voc["k"], voc["L"], voc["scoring"], voc["weighting"], voc["parent"], voc["desc"], voc["weight"], voc["is_leaf"])
    tk, tf = kb.transform(kd, 4), kb.transform(fd, 4)
    prob = dict(desc_kf=kd, desc_f=fd, kf_has_mp=(tc.last_mp[b, :len(kk)] >= 0).astype(np.uint8), angle_kf=kk["angle"], angle_f=fa,
                node_id_kf=tk["fv_node"], node_off_kf=tk["fv_off"], node_idx_kf=tk["fv_idx"], node_id_f=tf["fv_node"],
                node_off_f=tf["fv_off"], node_idx_f=tf["fv_idx"])
    n, m = pkg.Matcher(0.7, True).SearchByBoW(prob)
    assert n == res[b][0] and (m == res[b][1]).all()


def test_chain_on_recorded_pairs_vs_oracle(pkg, oracle, gpu, tmp_path):
    """the optional real-data path: (LastFrame, CurrentFrame) pairs read from a TUM-format directory (PNG decoding, timestamp
    association, depth in metres, TUM1.yaml's distortion) go through the same chain and equal the oracle like the generated ones"""
    sys.path.insert(0, os.path.dirname(oracle.__file__))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import parity
    from test_datasets_cpu import _mini_tum
    _mini_tum(pkg, tmp_path, n=5)
    pr = pkg.datasets.tum_pairs(str(tmp_path), 4)
    scen = pkg.scenario.tracking_scenario_real(pr, 8)
    scen["dist"] = np.asarray(TUM1_DIST, np.float32)
    tc = pkg.chain.TrackingChain(scen, n_local=1200)
    tc.step()
    tc.wait()
    co = parity.ChainOracle(scen, tc)
    assert parity.chain_mismatches(parity.chain_snapshot(pkg, tc), co, range(8)) == []
    nm = tc.d_nm.cpu().numpy()
    assert (nm[0] > 50).all() and (nm[3] > 30).all(), nm


def test_keyframe_work_vs_oracle(pkg, oracle, gpu):
    """LocalMapping's matcher work for a new keyframe on device-resident keyframes (src/LocalMapping.cc:272, 493):
    SearchForTriangulation(keyframe, neighbour, F12, ...) and the search part of Fuse(neighbour, the keyframe's map points) for every
    (keyframe, neighbour) pair in one call each -- vMatches12, counts, best_idx / best_dist equal the oracle's on its own extraction
    of the same images; matches exist (the neighbours see the same scene)."""
    sys.path.insert(0, os.path.dirname(oracle.__file__))
    import parity
    scen = pkg.scenario.tracking_scenario(31, 6, n_unique=3)
    tc = pkg.chain.TrackingChain(scen, n_local=800)
    voc = pkg.synth.synth_vocabulary(402, 10, 4)
    kw = pkg.chain.KeyFrameWork(tc, voc, n_kf=5, n_nb=4)
    co = parity.ChainOracle(scen, tc)
    for _ in range(2):
        kw.run()
        assert parity.keyframe_work_mismatches(kw, co, voc, range(len(kw.kf1))) == []
    assert (kw.nm > 20).all(), kw.nm
    assert ((kw.best_idx >= 0).sum(1) > 50).all()
    # with second-order neighbours (SearchInNeighbors, src/LocalMapping.cc:475-485): Fuse targets only -- 3 first-order neighbours are
    # triangulated against, 3 + 3 x 2 = 9 keyframes per keyframe are fused into
    # (and with the calls returning complete, the default of the C ABI: kw above enqueues its three calls and waits once)
    kw2 = pkg.chain.KeyFrameWork(tc, voc, n_kf=4, n_nb=3, n_second=2, async_calls=False)
    assert len(kw2.kf1) == 4 * 9 and len(kw2.tri_pairs) == 4 * 3 and (kw2.tri_of >= 0).sum() == 12
    kw2.run()
    co2 = parity.ChainOracle(scen, tc)   # (its cache of the neighbours' oracle extraction is keyed by neighbour index)
    assert parity.keyframe_work_mismatches(kw2.snapshot(), co2, voc, range(len(kw2.kf1))) == []
    assert kw2.match12.shape[0] == 12 and kw2.best_idx.shape[0] == 36 and ((kw2.best_idx >= 0).sum(1) > 50).all()
    # pairs that name keyframes outside the batches are rejected before anything runs
    with pytest.raises(pkg.AosError):
        tc.last.SearchForTriangulation(kw.kfs, [0], [10 ** 6], kw.F12[:1], kw.epipole[:1], kw.fv1[8].data_ptr(),
                                       [kw.fv1[k].data_ptr() for k in (3, 4, 5, 6)], [kw.fv2[k].data_ptr() for k in (3, 4, 5, 6)],
                                       kw.d_match12.data_ptr(), kw.d_nm.data_ptr())


def test_stereo_frame_chain_vs_oracle(pkg, oracle, gpu):
    """The stereo Frame constructor on the device (src/Frame.cc:57-113: both eyes extracted on two handles, ComputeStereoMatches
    enqueued behind both without a host wait, aos2_frames_build_stereo) and the tracking chain behind it, KITTI geometry (2000
    features: frames beyond 1536 queries take the 1024-thread greedy resolve), against the oracle chain whose Frame members come from
    the oracle's ComputeStereoMatches: mvuRight / mvDepth and every later member bit for bit, poses 1e-5."""
    sys.path.insert(0, os.path.dirname(oracle.__file__))
    import parity
    scen = pkg.scenario.tracking_scenario(41, 3, cfg="kitti", n_unique=3, stereo=True)
    tc = pkg.chain.StereoTrackingChain(scen, n_local=1500)
    assert tc.cap > 1536
    tc.step()
    tc.wait()
    co = parity.ChainOracle(scen, tc, th_last=tc.th_last, th_local=tc.th_local, nnratio_local=tc.nnratio_local)
    snap = parity.chain_snapshot(pkg, tc)
    assert parity.chain_mismatches(snap, co, range(3)) == []
    assert (snap["depth"][:, :1500] > 0).sum() > 1500 and (snap["nm"][0] > 300).all()   # stereo matches exist, the chain tracks
    # the synchronous ComputeStereoMatches call on the same extractions gives the same arrays
    import torch
    ur, dp = torch.zeros_like(tc.d_ur), torch.zeros_like(tc.d_dp)
    pkg.capi.compute_stereo_matches_device(tc.ex, tc.ex_r, tc.B, tc.d_kps.data_ptr(), tc.d_desc.data_ptr(), tc.d_n.data_ptr(), tc.r_kps.data_ptr(),
                                           tc.r_desc.data_ptr(), tc.r_n.data_ptr(), tc.cap, tc.mb, tc.mbf, ur.data_ptr(), dp.data_ptr())
    assert torch.equal(ur, tc.d_ur) and torch.equal(dp, tc.d_dp)


def test_native_step_runner_replays_the_recorded_jobs(pkg, oracle, gpu):
    """bench.py's step schedule on native threads (csrc/host_runner.cpp): the C calls of the tracking chain, the keyframe legs and a
    LocalBA batch are recorded once (capi.recording: addresses + arguments, nothing executes) and replayed by the runner -- the chain
    on the stepping thread, the other two on a thread each.  What the replays leave behind equals the oracle's results like the
    directly called forms do, step after step, with two pipelines in flight; a failing call stops the runner with its status."""
    sys.path.insert(0, os.path.dirname(oracle.__file__))
    import parity
    capi = pkg.capi
    scen = pkg.scenario.tracking_scenario(41, 12, n_unique=4)
    voc = pkg.synth.synth_vocabulary(403, 10, 4)
    pipes = [pkg.chain.TrackingChain(scen, n_local=800) for _ in range(2)]
    bows = [pkg.chain.ReferenceKeyFrameBoW(p, voc, 6) for p in pipes]
    kfws = [pkg.chain.KeyFrameWork(p, voc, n_kf=3, n_nb=3) for p in pipes]
    probs = [pkg.synth.synth_lba_problem(90 + i, n_local=5 + 3 * i, n_fixed=3, n_points=300) for i in range(3)]
    lba = pkg.LocalBA()
    prep = lba.prepare_batch(probs)
    r = capi.Runner(2, 1)
    for j, p in enumerate(pipes):
        with capi.recording() as c:
            p.wait()
        assert len(c) == 2
        r.set_list(r.PIPE_WAIT, j, c)
        with capi.recording() as c:
            p.step()
            bows[j].order()
        assert len(c) == 9   # nothing ran: the batch has no members yet
        assert p.cur.device_ptr(capi.Frames.TCW) == 0
        r.set_list(r.PIPE_STEP, j, c)
        with capi.recording() as c:
            bows[j].run_calls()
            kfws[j].run_calls()
        r.set_list(r.KF_JOB, j, c)
    with capi.recording() as c:
        lba.solve_prepared(prep)
    assert len(c) == 1
    r.set_list(r.LBA_JOB, 0, c)
    co = [parity.ChainOracle(scen, p) for p in pipes]
    want = [oracle.lba_solve(q) for q in probs]
    r.run(0, 5)
    r.sync()
    for j, p in enumerate(pipes):
        assert parity.chain_mismatches(parity.chain_snapshot(pkg, p), co[j], range(12)) == []
        assert parity.bow_leg_mismatches(bows[j].get_results(), co[j], voc, range(6)) == []
        assert parity.keyframe_work_mismatches(kfws[j].snapshot(), co[j], voc, range(len(kfws[j].kf1))) == []
    for w, q in enumerate(probs):
        got = pkg.LocalBA._result(prep["R"][w], prep["arrs"][w])
        assert parity.lba_mismatches(got, want[w], tag=f"window {w}") == []
    assert len(r.stats(3)) == 5 and len(r.stats(1)) == 5 and len(r.stats(2)) == 5
    # a call that fails stops the schedule and names itself
    with capi.recording() as c:
        capi.lib().aos2_frames_wait(None)
    r.set_list(r.PIPE_STEP, 0, c)
    with pytest.raises(pkg.AosError) as ei:
        r.step(6)
    assert "kind 1 index 0 call 0" in str(ei.value)


def test_host_boundary_of_the_chain(pkg, oracle, gpu):
    """bench.py --host-images: the images of every step come from page-locked host memory (one copy up on a stream of the harness, the
    extraction ordered behind it on the device: aos2_extractor_wait_for_stream) and the Frame's members land in page-locked host
    arrays behind the second PoseOptimization.  The host arrays hold what the device holds, and what the oracle computes -- also
    after the device copy of the images was scribbled over between the steps (the chain really reads the host images)."""
    sys.path.insert(0, os.path.dirname(oracle.__file__))
    import parity
    for stereo, prefetch in ((False, False), (True, False), (False, True), (True, True)):   # prefetch: uploads one step ahead, via a staging buffer
        scen = pkg.scenario.tracking_scenario(43, 6, cfg="kitti" if stereo else "tum", n_unique=3, stereo=stereo)
        tc = (pkg.chain.StereoTrackingChain if stereo else pkg.chain.TrackingChain)(scen, n_local=800)
        tc.step()
        tc.wait()
        tc.enable_host_boundary(prefetch=prefetch)
        co = parity.ChainOracle(scen, tc)
        for _ in range(2):
            tc.d_cur.zero_()
            if stereo:
                tc.d_right.zero_()
            tc.torch.cuda.synchronize()
            tc.step()
            tc.wait()
            host, dev = parity.chain_snapshot_host(pkg, tc), parity.chain_snapshot(pkg, tc)
            for k in ("n", "kps", "desc", "mp", "outlier", "Tcw", "u_right", "depth", "nm"):
                assert host[k].tobytes() == dev[k].tobytes(), k
            assert parity.chain_mismatches(host, co, range(6)) == []
        assert tc.hb["up_bytes"] == (2 if stereo else 1) * 6 * scen["w"] * scen["h"] and tc.hb["down_bytes"] > 6 * tc.cap * 60


def test_recorded_step_replays_as_one_launch(pkg, oracle, gpu):
    """include/aos2.h "Replay of a fixed call sequence": the calls of the chain's step recorded once (aos2_capture_begin / _end on the
    Frame batch's stream, the extractors' streams joining through aos2_extractor_wait_for_stream), then one aos2_graph_launch per
    frame.  After the outputs were scribbled over and the images replaced, the replay leaves what the plain calls leave, and what
    the oracle computes; a call that waits on the host inside a recording is reported, not recorded."""
    sys.path.insert(0, os.path.dirname(oracle.__file__))
    import parity
    keys = ("n", "kps", "desc", "mp", "outlier", "Tcw", "u_right", "depth", "nm")
    for stereo, nf in ((False, 1), (False, 6), (True, 3)):
        scen = pkg.scenario.tracking_scenario(47, nf, cfg="kitti" if stereo else "tum", n_unique=min(nf, 3), stereo=stereo)
        tc = (pkg.chain.StereoTrackingChain if stereo else pkg.chain.TrackingChain)(scen, n_local=800)
        tc.step()
        tc.wait()
        want = parity.chain_snapshot(pkg, tc)
        co = parity.ChainOracle(scen, tc)
        gr = tc.capture_step()
        assert gr.nodes() >= 15
        img = tc.d_cur.clone()
        for rep in range(3):
            tc.d_kps.zero_(); tc.d_desc.zero_(); tc.d_n.zero_(); tc.d_nm.zero_()
            tc.d_cur.zero_()
            tc.torch.cuda.synchronize()
            tc.d_cur.copy_(img)   # "the next image" arrives in the same buffer
            tc.torch.cuda.synchronize()
            tc.step_graph()
            tc.cur.wait()
            got = parity.chain_snapshot(pkg, tc)
            for k in keys:
                assert got[k].tobytes() == want[k].tobytes(), (stereo, nf, rep, k)
            assert parity.chain_mismatches(got, co, range(nf)) == []
        tc.wait()   # the extractor's own wait reports the replayed extraction's status
        tc.step(); tc.wait()   # the plain calls still work on the same handles
        got = parity.chain_snapshot(pkg, tc)
        assert all(got[k].tobytes() == want[k].tobytes() for k in keys)
        gr.close()
    # a host wait inside a recording is refused (before the runtime sees it), the handles stay usable
    with pytest.raises(RuntimeError) as ei:
        with pkg.capi.Graph.capture(tc.cur.stream()):
            tc.cur.wait()
    assert "recording" in str(ei.value)
    tc.step(); tc.wait()
    got = parity.chain_snapshot(pkg, tc)
    assert all(got[k].tobytes() == want[k].tobytes() for k in keys)

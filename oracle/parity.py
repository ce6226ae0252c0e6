"""ORACLE (test infrastructure): comparisons of the HIP path's results with the oracle's on the SAME inputs, shared by
tests/ and by bench.py's --verify leg (which checks the last TIMED step).  Only tests/, smoke() and bench.py may use it.

Every function returns a list of mismatch descriptions (empty = parity holds) instead of asserting, so that bench.py can
report them in its JSON line and the tests can assert on the list.

Bars (north_star): keypoints, descriptors, Frame members, map point assignments, outlier flags and
counts bit-identical; poses / points within 1e-5 absolute of the oracle's float32 write-back (close()); identical Levenberg-Marquardt iteration and trial counts.
"""
from __future__ import annotations

import numpy as np

import oracle as O
import chain as ochain

TOL = 1e-5          # north_star's bar for poses / points, absolute, literally
WORST = {}          # worst |difference| seen per quantity since reset_worst() (reported by bench.py / the tests' messages)


def reset_worst():
    WORST.clear()


def close(a, b, tol=TOL, key=None):
    """|a - b| <= 1e-5 everywhere.  The one exception is a float32 value of magnitude >= 128: there the spacing of the float32
    the reference writes back (Converter::toCvMat, Optimizer.cc:763-778) is 1.5e-5 or more, so ONE float32 step is the
    resolution of the stored value -- allowed there and nowhere else (no synthetic case of this repo reaches it: the worst
    observed differences are recorded in WORST and reported)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return True
    d = np.abs(a - b)
    if key is not None:
        WORST[key] = max(WORST.get(key, 0.0), float(d.max()))
    big = np.abs(b) >= 128.0
    ok = d <= tol
    if big.any():
        ok = ok | (big & (d <= np.spacing(np.abs(b).astype(np.float32)).astype(np.float64)))
    return bool(ok.all())


def worst(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()) if np.size(a) else 0.0


class ChainOracle:
    """The oracle's results for the distinct (LastFrame, CurrentFrame) pairs of a scenario.tracking_scenario(), computed on
    demand and kept (a tiled batch repeats them).  `tc` = the TrackingChain whose LastFrame members / map were built from
    the scenario (chain.py builds them on the host from the extractor's keypoints of the last / older views)."""

    def __init__(self, scen, tc, th_last=15.0, th_local=3.0, nnratio_local=0.8):
        self.scen, self.tc = scen, tc
        self.oe = O.Extractor(nfeatures=scen["nfeatures"])
        self.oe_r = O.Extractor(nfeatures=scen["nfeatures"]) if scen.get("stereo") else None
        self.th = (th_last, th_local, nnratio_local)
        self.cache = {}

    def unique(self, u, timing=None):
        if u in self.cache and timing is None:
            return self.cache[u]
        import time
        scen, tc = self.scen, self.tc
        t0 = time.perf_counter()
        if self.oe_r is not None:
            # the stereo Frame constructor: the two eyes are extracted by two threads (src/Frame.cc:103-109), then ComputeStereoMatches
            import threading
            res = {}
            thr = threading.Thread(target=lambda: res.__setitem__("r", self.oe_r.extract(scen["right_cur"][u])))
            thr.start()
            okps, odesc = self.oe.extract(scen["cur"][u])
            thr.join()
            t1 = time.perf_counter()
            f = ochain.frame_from_stereo(self.oe, self.oe_r, okps, odesc, res["r"][0], res["r"][1], scen, self.oe.scale_factors, self.oe.inv_sigma2)
        else:
            okps, odesc = self.oe.extract(scen["cur"][u])
            t1 = time.perf_counter()
            f = ochain.frame_from_extraction(okps, odesc, scen["depth_cur"][u], scen, self.oe.scale_factors, self.oe.inv_sigma2)
        t2 = time.perf_counter()
        if timing is not None:
            timing["extract"] = timing.get("extract", 0.0) + t1 - t0
            timing["frame_build"] = timing.get("frame_build", 0.0) + t2 - t1
        lk = tc.host_last[u][0]
        last_h = dict(mp=tc.last_mp[u, :len(lk)], outlier=tc.last_outlier[u, :len(lk)], kp_octave=lk["octave"], kp_angle=lk["angle"])
        out = ochain.track_frame(f, last_h, tc.map["table"], tc.map["local"][u], scen["Tcw_guess"][u], scen["Tlw"][u], scen,
                                 th_last=self.th[0], th_local=self.th[1], nnratio_local=self.th[2], timing=timing)
        self.cache[u] = (okps, odesc, f, out)
        return self.cache[u]


def _precompute(self, us, threads=8):
    """fill the cache for the distinct pairs `us` on `threads` host threads (one oracle extractor per thread; the C
    functions hold no state and release the GIL)"""
    todo = [u for u in us if u not in self.cache]
    if threads <= 1 or len(todo) < 4:
        for u in todo:
            self.unique(u)
        return
    import threading
    from concurrent.futures import ThreadPoolExecutor
    tl = threading.local()

    def work(u):
        co = getattr(tl, "co", None)
        if co is None:
            co = tl.co = ChainOracle(self.scen, self.tc, *self.th)
            co.cache = self.cache
        co.unique(u)
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, todo))


ChainOracle.precompute = _precompute


def chain_snapshot(pkg, tc):
    """host copies of what TrackingChain.step() + wait() left on the device (taken before anything else reuses the batch)"""
    F = pkg.capi.Frames
    B, cap = tc.B, tc.cap
    return dict(B=B, cap=cap, n=tc.d_n.cpu().numpy(),
                kps=tc.d_kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28).copy().view(pkg.capi.KP_DTYPE).reshape(B, cap),
                desc=tc.d_desc.cpu().numpy(), mp=tc.cur.get(F.MAP_POINTS), outlier=tc.cur.get(F.OUTLIER), Tcw=tc.cur.get(F.TCW),
                u_right=tc.cur.get(F.U_RIGHT), depth=tc.cur.get(F.DEPTH), nm=tc.d_nm.cpu().numpy())


def chain_snapshot_host(pkg, tc):
    """the same members from the page-locked HOST arrays a TrackingChain with enable_host_boundary() fills every step"""
    hb, B, cap = tc.hb, tc.B, tc.cap
    d = {k: h.numpy().copy() for k, _, h in hb["down"] + hb["members"]}
    d["kps"] = d["kps"].view(np.uint8).reshape(B, cap, 28).copy().view(pkg.capi.KP_DTYPE).reshape(B, cap)
    d.update(B=B, cap=cap)
    return d


def chain_mismatches(snap, co: ChainOracle, positions):
    """The device-resident chain's members (chain_snapshot) against the oracle chain, for the batch positions given
    (position b holds unique pair scen["index"][b])."""
    scen = co.scen
    n, kps, desc, mp, outl, T, ur, dp, nm = (snap[k] for k in ("n", "kps", "desc", "mp", "outlier", "Tcw", "u_right", "depth", "nm"))
    bad = []
    for b in positions:
        u = int(scen["index"][b])
        okps, odesc, f, w = co.unique(u)
        tag = f"frame {b} (pair {u})"
        if n[b] != len(okps):
            bad.append(f"{tag}: {n[b]} keypoints, oracle {len(okps)}")
            continue
        k = int(n[b])
        if kps[b, :k].tobytes() != okps.tobytes():
            bad.append(f"{tag}: keypoints differ")
        if not (desc[b, :k] == odesc).all():
            bad.append(f"{tag}: descriptors differ")
        if ur[b, :k].tobytes() != f["u_right"].tobytes() or dp[b, :k].tobytes() != f["depth"].tobytes():
            bad.append(f"{tag}: mvuRight / mvDepth differ")
        want_nm = (w["nmatches_last"], w["inliers_1"], w["nmatches_local"], w["inliers_2"])
        if tuple(int(v) for v in nm[:, b]) != tuple(int(v) for v in want_nm):
            bad.append(f"{tag}: counts {tuple(int(v) for v in nm[:, b])}, oracle {tuple(int(v) for v in want_nm)}")
        if not (mp[b, :k] == w["mp_after_local"]).all() or not (mp[b, k:] == -1).all():
            bad.append(f"{tag}: mvpMapPoints differ in {int((mp[b, :k] != w['mp_after_local']).sum())} features")
        if not (outl[b, :k] == w["outlier_2"]).all():
            bad.append(f"{tag}: mvbOutlier differs in {int((outl[b, :k] != w['outlier_2']).sum())} features")
        if not close(T[b], w["Tcw_2"], key="mTcw"):
            bad.append(f"{tag}: mTcw off by {worst(T[b], w['Tcw_2']):.3g}")
    return bad


# Re-associations of the oracle's own arithmetic (lba_oracle.c orc_set_lba_variant, oracle/Makefile "contracted"): (edges
# reversed, variant bits, contracted build).  Every one computes each quantity of the default run to a few units of the last place.
LBA_REASSOCIATIONS = (("a edges reversed", True, 0, False), ("b reduced system eliminated in reverse", False, 1, False),
                      ("c long double accumulation", False, 2, False), ("d b + c", False, 3, False),
                      ("e built with g2o's flags (fused multiply-adds)", False, 0, True), ("f e + b", False, 1, True))
# (the six re-associations are a SAMPLE of what rounding can do to the window; over 192 windows started off the optimum the device's
# difference was at most 1.79 x the sample's spread, median 0, 90 % 0.56 x: profiles/r06_lba_offopt_sweep.txt -- 4 leaves a factor of two)
LBA_RESOLUTION_FACTOR = 4.0


def lba_resolution(prob, want=None, iters=(5, 10)):
    """How far apart do faithful executions of LocalBundleAdjustment end on THIS window?  The oracle against itself under
    LBA_REASSOCIATIONS: {"pose": max |difference| of the float32 Tcw entries over the runs, "point": ... of the points,
    "decisions_equal": iterations, trials and outlier sets of every run equal the default run's, "rows": per run}.
    A window that starts near the optimum converges and the spread is 0 (below float32); one that starts far off (15 iterations
    do not converge, landmarks left with two inlier observations are held by lambda alone) amplifies rounding-level differences to
    1e-5 .. 1e-3 -- measured in profiles/r06_lba_sensitivity.txt.  That spread is the resolution at which ANY implementation can be
    compared with the oracle on such a window; lba_mismatches(resolution=...) allows LBA_RESOLUTION_FACTOR times it (never less than
    1e-5)."""
    w0 = want if want is not None else O.lba_solve(prob, iters1=iters[0], iters2=iters[1])
    rows, pose, point, same = [], 0.0, 0.0, True
    for name, rev, bits, con in LBA_REASSOCIATIONS:
        q, back = prob, None
        if rev:
            order = np.arange(prob["n_edges"])[::-1].copy()
            q = dict(prob)
            for k in ("edge_pose", "edge_point", "edge_obs", "edge_stereo", "edge_inv_sigma2"):
                q[k] = np.ascontiguousarray(prob[k][order])
            back = np.argsort(order)
        w1 = O.lba_solve(q, iters1=iters[0], iters2=iters[1], variant=bits, contracted=con)
        out1 = w1["edge_outlier"][back] if rev else w1["edge_outlier"]
        eq = tuple(w0["iters"]) == tuple(w1["iters"]) and w0["trials"] == w1["trials"] and bool((w0["edge_outlier"] == out1).all())
        dp, dx = worst(w0["pose_Tcw"], w1["pose_Tcw"]), worst(w0["point_xyz"], w1["point_xyz"])
        rows.append(dict(name=name, pose=dp, point=dx, decisions_equal=eq, outlier_flags_differ=int((w0["edge_outlier"] != out1).sum()),
                         pose64=worst(w0["pose_qt"], w1["pose_qt"]), point64=worst(w0["point_xyz64"], w1["point_xyz64"])))
        pose, point, same = max(pose, dp), max(point, dx), same and eq
    return dict(pose=pose, point=point, decisions_equal=same, rows=rows)


def lba_mismatches(got, want, tag="window", resolution=None):
    """one LocalBundleAdjustment result (capi.LocalBA._result dict) against oracle.lba_solve's.  resolution = lba_resolution(prob)
    for a window that starts off the optimum: poses / points then within max(1e-5, LBA_RESOLUTION_FACTOR x the oracle's own spread
    on that window) and the final chi2 within the same factor of 1e-6; every decision still has to be the oracle's."""
    bad = []
    if resolution is not None:
        tp, tx = max(TOL, LBA_RESOLUTION_FACTOR * resolution["pose"]), max(TOL, LBA_RESOLUTION_FACTOR * resolution["point"])
        if got["status"] != 0:
            bad.append(f"{tag}: status {got['status']}")
        if tuple(got["iters"]) != tuple(want["iters"]):
            bad.append(f"{tag}: iterations {tuple(got['iters'])}, oracle {tuple(want['iters'])}")
        if sum(got["trials"]) != want["trials"]:
            bad.append(f"{tag}: {sum(got['trials'])} LM trials, oracle {want['trials']}")
        dp, dx = worst(got["pose_Tcw"], want["pose_Tcw"]), worst(got["point_xyz"], want["point_xyz"])
        WORST["lba_offopt_pose"] = max(WORST.get("lba_offopt_pose", 0.0), dp)
        WORST["lba_offopt_point"] = max(WORST.get("lba_offopt_point", 0.0), dx)
        if dp > tp:
            bad.append(f"{tag}: poses off by {dp:.3g}, allowed {tp:.3g} (oracle's own spread {resolution['pose']:.3g})")
        if dx > tx:
            bad.append(f"{tag}: points off by {dx:.3g}, allowed {tx:.3g} (oracle's own spread {resolution['point']:.3g})")
        if not (got["edge_outlier"] == want["edge_outlier"]).all():
            bad.append(f"{tag}: outlier sets differ in {int((got['edge_outlier'] != want['edge_outlier']).sum())} edges")
        return bad
    if got["status"] != 0:
        bad.append(f"{tag}: status {got['status']}")
    if tuple(got["iters"]) != tuple(want["iters"]):
        bad.append(f"{tag}: iterations {tuple(got['iters'])}, oracle {tuple(want['iters'])}")
    if sum(got["trials"]) != want["trials"]:
        bad.append(f"{tag}: {sum(got['trials'])} LM trials, oracle {want['trials']}")
    if not close(got["pose_Tcw"], want["pose_Tcw"], key="lba_pose"):
        bad.append(f"{tag}: poses off by {worst(got['pose_Tcw'], want['pose_Tcw']):.3g}")
    if not close(got["point_xyz"], want["point_xyz"], key="lba_point"):
        bad.append(f"{tag}: points off by {worst(got['point_xyz'], want['point_xyz']):.3g}")
    if not (got["edge_outlier"] == want["edge_outlier"]).all():
        bad.append(f"{tag}: outlier sets differ in {int((got['edge_outlier'] != want['edge_outlier']).sum())} edges")
    c = want["chi2_trace"][-1] if len(want["chi2_trace"]) else 0.0
    WORST["lba_final_chi2_rel"] = max(WORST.get("lba_final_chi2_rel", 0.0), abs(got["final_chi2"] - c) / max(c, 1e-30))
    if abs(got["final_chi2"] - c) > 1e-6 * max(c, 1e-30):
        bad.append(f"{tag}: final chi2 {got['final_chi2']!r}, oracle {c!r}")
    return bad


def stereo_slot_mismatches(pkg, kps, desc, n, u_right, depth, left_img, right_img, cfg, tag="frame"):
    """One stereo frame of the EuRoC workload (both eyes' ORBextractor::operator() + Frame::ComputeStereoMatches,
    src/Frame.cc:495-669): left keypoints / descriptors and mvuRight / mvDepth against the oracle, bit for bit."""
    bad = []
    oe_l, oe_r = O.Extractor(nfeatures=cfg["nfeatures"]), O.Extractor(nfeatures=cfg["nfeatures"])
    okl, odl = oe_l.extract(left_img)
    okr, odr = oe_r.extract(right_img)
    if n != len(okl):
        return [f"{tag}: {n} keypoints, oracle {len(okl)}"]
    if np.ascontiguousarray(kps[:n]).tobytes() != okl.tobytes():
        bad.append(f"{tag}: left keypoints differ")
    if not (desc[:n] == odl).all():
        bad.append(f"{tag}: left descriptors differ")
    mbf = np.float32(cfg["bf"])
    mb = np.float32(mbf / np.float32(cfg["fx"]))
    our, odp, _ = O.compute_stereo_matches(oe_l, oe_r, okl, odl, okr, odr, mb, mbf)
    if u_right[:n].tobytes() != our.tobytes() or depth[:n].tobytes() != odp.tobytes():
        bad.append(f"{tag}: mvuRight / mvDepth differ in {int((u_right[:n] != our).sum())} features")
    return bad


def bow_leg_mismatches(results, co: ChainOracle, voc: dict, positions, nnratio=0.7, levelsup=4, timing=None):
    """chain.ReferenceKeyFrameBoW's results (Frame::ComputeBoW + SearchByBoW(reference keyframe, frame), src/Tracking.cc:858-866)
    for the batch positions given, against the oracle's vocabulary transform and SearchByBoW on the oracle's own extraction
    of the same images: match arrays and counts bit-identical."""
    import time
    tc, scen = co.tc, co.scen
    if not hasattr(co, "_ov"):
        co._ov = O.Vocabulary()
        co._ov.set_nodes(voc["k"], voc["L"], voc["scoring"], voc["weighting"], voc["parent"], voc["desc"], voc["weight"], voc["is_leaf"])
        co._kf_bow = {}
    bad = []
    for b in positions:
        u = int(scen["index"][b])
        okps, odesc, f, w = co.unique(u)
        lk, ld = tc.host_last[u]
        if u not in co._kf_bow:
            co._kf_bow[u] = co._ov.transform(ld, levelsup)   # KeyFrame::ComputeBoW, once per keyframe
        t0 = time.perf_counter()
        fb = co._ov.transform(odesc, levelsup)
        t1 = time.perf_counter()
        kb = co._kf_bow[u]
        prob = dict(desc_kf=ld, desc_f=odesc, kf_has_mp=(tc.last_mp[u, :len(lk)] >= 0).astype(np.uint8), angle_kf=lk["angle"],
                    angle_f=okps["angle"], node_id_kf=kb["fv_node"], node_off_kf=kb["fv_off"], node_idx_kf=kb["fv_idx"],
                    node_id_f=fb["fv_node"], node_off_f=fb["fv_off"], node_idx_f=fb["fv_idx"], nnratio=np.float32(nnratio),
                    check_orientation=1)
        n, m = O.search_by_bow(prob)
        if timing is not None:
            timing["compute_bow"] = timing.get("compute_bow", 0.0) + t1 - t0
            timing["search_by_bow"] = timing.get("search_by_bow", 0.0) + time.perf_counter() - t1
        if results is None:
            continue
        gn, gm = results[b]
        if gn != n or len(gm) != len(m) or not (np.asarray(gm) == m).all():
            bad.append(f"frame {b} (pair {u}): SearchByBoW {gn} matches, oracle {n}" + ("" if len(gm) != len(m) else f", {int((np.asarray(gm) != m).sum())} entries differ"))
    return bad


def keyframe_work_mismatches(kw, co: ChainOracle, voc: dict, pairs, levelsup=None, timing=None):
    """chain.KeyFrameWork's results (SearchForTriangulation + the search part of Fuse for (keyframe, neighbour) pairs,
    src/LocalMapping.cc:272, 493) for the pair indices given, against the oracle on its own extraction of the same images:
    vMatches12 / counts and Fuse's best_idx / best_dist bit-identical.  kw = None: only run (and time) the oracle side."""
    import time
    tc, scen = co.tc, co.scen
    if not hasattr(co, "_ov"):
        co._ov = O.Vocabulary()
        co._ov.set_nodes(voc["k"], voc["L"], voc["scoring"], voc["weighting"], voc["parent"], voc["desc"], voc["weight"], voc["is_leaf"])
        co._kf_bow = {}
    if not hasattr(co, "_nb"):
        co._nb, co._kf1 = {}, {}
    sf, isg = co.oe.scale_factors, co.oe.inv_sigma2
    table = tc.map["table"]
    bad = []
    src = kw if kw is not None else co._kw_inputs
    levelsup = int(getattr(src, "levelsup", 4)) if levelsup is None else levelsup
    only_stereo, check_ori = int(getattr(src, "only_stereo", False)), int(getattr(src, "check_orientation", False))
    for p in pairs:
        b, j = int(src.kf1[p]), int(src.kf2[p])
        u = int(scen["index"][b])
        ts = time.perf_counter()
        if u not in co._kf1:   # keyframe 1 = the scene's LastFrame view
            lk, ld = tc.host_last[u]
            f1 = ochain.frame_from_extraction(lk, ld, scen["depth_last"][u], scen, sf, isg)
            co._kf1[u] = (lk, ld, f1, co._ov.transform(ld, levelsup))
        if j not in co._nb:    # the neighbour keyframe: the oracle's own extraction + Frame members + BoW
            okps, odesc = co.oe.extract(src.nb["imgs"][j])
            depth = np.full((scen["h"], scen["w"]), np.float32(scen["Z"][j // src.n_nb]), np.float32)
            f2 = ochain.frame_from_extraction(okps, odesc, depth, scen, sf, isg)
            co._nb[j] = (okps, odesc, f2, co._ov.transform(odesc, levelsup))
        if timing is not None:   # (the keyframes exist before the step: their extraction is not part of it)
            timing["_setup"] = timing.get("_setup", 0.0) + time.perf_counter() - ts
        lk, ld, f1, b1 = co._kf1[u]
        okps, odesc, f2, b2 = co._nb[j]
        n1, n2 = len(lk), len(okps)
        t0 = time.perf_counter()
        prob = dict(desc1=ld, desc2=odesc, has_mp1=(tc.last_mp[b, :n1] >= 0).astype(np.uint8), has_mp2=(src.nb_mp[j, :n2] >= 0).astype(np.uint8),
                    x1=f1["kp_x"], y1=f1["kp_y"], angle1=np.ascontiguousarray(lk["angle"], np.float32), u_right1=f1["u_right"],
                    x2=f2["kp_x"], y2=f2["kp_y"], angle2=np.ascontiguousarray(okps["angle"], np.float32), u_right2=f2["u_right"],
                    octave2=np.ascontiguousarray(okps["octave"], np.int32), scale_factors2=np.ascontiguousarray(sf, np.float32),
                    level_sigma2_2=(np.asarray(sf, np.float32) * np.asarray(sf, np.float32)).astype(np.float32), F12=src.F12[p],
                    ex=np.float32(src.epipole[p, 0]), ey=np.float32(src.epipole[p, 1]), only_stereo=only_stereo, check_orientation=check_ori,
                    node_id1=b1["fv_node"], node_off1=b1["fv_off"], node_idx1=b1["fv_idx"], node_id2=b2["fv_node"], node_off2=b2["fv_off"],
                    node_idx2=b2["fv_idx"])
        tri = int(src.tri_of[p]) if hasattr(src, "tri_of") else p   # (second-order neighbours are fuse targets only: no triangulation)
        n, m = O.search_for_triangulation(prob) if tri >= 0 else (0, None)
        t1 = time.perf_counter()
        # Fuse(pKFi = the neighbour, vpMapPointMatches of keyframe 1): the candidates are table rows
        rows = src.fuse_rows[p]
        pts = _fuse_points(rows, src.nb["Tkw"][j], table, scen, f2, src.fuse_th)
        nf, bi, bd = O.fuse(f2, pts, sim3=False)
        t2 = time.perf_counter()
        if timing is not None:
            timing["search_for_triangulation"] = timing.get("search_for_triangulation", 0.0) + t1 - t0
            timing["fuse"] = timing.get("fuse", 0.0) + t2 - t1
        if kw is None:
            continue
        tag = f"keyframe {b} / neighbour {j}"
        if tri >= 0 and (int(kw.nm[tri]) != n or not (kw.match12[tri, :n1] == m).all() or not (kw.match12[tri, n1:] == -1).all()):
            bad.append(f"{tag}: SearchForTriangulation {int(kw.nm[tri])} matches, oracle {n}; {int((kw.match12[tri, :n1] != m).sum())} entries differ")
        if not (kw.best_idx[p] == bi).all() or not (kw.best_dist[p] == bd).all():
            bad.append(f"{tag}: Fuse best_idx differs in {int((kw.best_idx[p] != bi).sum())} of {int((rows >= 0).sum())} candidates (oracle fuses {nf})")
    # Fuse(mpCurrentKeyFrame, vpFuseCandidates) (src/LocalMapping.cc:518) for the keyframes the pairs name: the candidates into the keyframe itself
    for b in sorted(set(int(src.kf1[p]) for p in pairs)):
        u = int(scen["index"][b])
        lk, ld, f1, b1 = co._kf1[u]
        t0 = time.perf_counter()
        pts = _fuse_points(src.rev_rows[b], scen["Tlw"][u], table, scen, f1, src.fuse_th)
        nf, bi, bd = O.fuse(f1, pts, sim3=False)
        if timing is not None:
            timing["fuse"] = timing.get("fuse", 0.0) + time.perf_counter() - t0
        if kw is not None and (not (kw.rev_idx[b] == bi).all() or not (kw.rev_dist[b] == bd).all()):
            bad.append(f"keyframe {b}: Fuse of the neighbourhood's points into it: best_idx differs in {int((kw.rev_idx[b] != bi).sum())} of "
                       f"{int((src.rev_rows[b] >= 0).sum())} candidates (oracle fuses {nf})")
    return bad


def _fuse_points(rows, Tcw, table, scen, f, th):
    """the candidate set of one Fuse(pKF, vpMapPoints, th) call as the oracle takes it: table rows (-1 = rejected by the loop head), pKF's pose"""
    T = np.asarray(Tcw, np.float32)
    R, t = np.ascontiguousarray(T[:3, :3]), np.ascontiguousarray(T[:3, 3])
    Rd, td = R.astype(np.float64), t.astype(np.float64)
    Ow = np.array([-((Rd[0, k] * td[0] + Rd[1, k] * td[1]) + Rd[2, k] * td[2]) for k in range(3)]).astype(np.float32)
    rr = np.where(rows >= 0, rows, 0)
    return dict(n_pts=len(rows), valid=(rows >= 0).astype(np.uint8), pos=table["pos"][rr], max_dist=table["max_dist"][rr],
                min_dist=table["min_dist"][rr], normal=table["normal"][rr], desc=table["desc"][rr], q_angle=np.zeros(len(rows), np.float32),
                R=R.reshape(9), t=t, Ow=Ow, R2=np.zeros(9, np.float32), t2=np.zeros(3, np.float32), fx=np.float32(scen["fx"]),
                fy=np.float32(scen["fy"]), cx=np.float32(scen["cx"]), cy=np.float32(scen["cy"]), bf=np.float32(scen["mbf"]),
                log_scale_factor=np.float32(np.log(np.float64(f["scale_factors"][1]))), inv_level_sigma2=f["inv_sigma2"], th=np.float32(th))

"""ORACLE (test infrastructure): the per-frame tracking chain on the host, step by step, from the oracle's C restatements
plus the index bookkeeping of Tracking.cc between them.  Only tests/, smoke() and bench.py's cpu_baseline may use it.

Chain (reference): Frame::Frame (src/Frame.cc:116-170) -> Tracking::TrackWithMotionModel (src/Tracking.cc:963-1027:
SearchByProjection(Current, Last), PoseOptimization, outlier discard) -> Tracking::SearchLocalPoints (:1302-1352:
isInFrustum + SearchByProjection(F, vpMapPoints)) -> PoseOptimization (Tracking::TrackLocalMap :1039).
"""
from __future__ import annotations

import time

import numpy as np

import oracle as O


def frame_from_extraction(kps, desc, depth_img, scen, scale_factors, inv_sigma2):
    """Frame::Frame after ExtractORB: mvKeysUn = mvKeys (no distortion), ComputeStereoFromRGBD, AssignFeaturesToGrid"""
    W, H = scen["w"], scen["h"]
    kx, ky = np.ascontiguousarray(kps["x"], np.float32), np.ascontiguousarray(kps["y"], np.float32)
    ux, uy = kx, ky
    min_x, max_x, min_y, max_y = np.float32(0), np.float32(W), np.float32(0), np.float32(H)
    dist = scen.get("dist")
    if dist is not None and dist[0] != 0:   # Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:433-493)
        K = (scen["fx"], scen["fy"], scen["cx"], scen["cy"])
        u = O.undistort_points(np.stack([kx, ky], 1), *K, dist)
        ux, uy = np.ascontiguousarray(u[:, 0]), np.ascontiguousarray(u[:, 1])
        c = O.undistort_points(np.array([[0, 0], [W, 0], [0, H], [W, H]], np.float32), *K, dist)
        min_x, max_x = min(c[0, 0], c[2, 0]), max(c[1, 0], c[3, 0])
        min_y, max_y = min(c[0, 1], c[1, 1]), max(c[2, 1], c[3, 1])
    ur, dp = O.stereo_from_rgbd(kx, ky, ux, depth_img, scen["mbf"])
    gwi, ghi = np.float32(64) / np.float32(max_x - min_x), np.float32(48) / np.float32(max_y - min_y)
    goff, gidx = O.assign_features_to_grid(ux, uy, min_x, min_y, gwi, ghi)
    gi = np.zeros(max(len(kx), 1), np.int32)
    gi[: len(gidx)] = gidx
    kx, ky = ux, uy   # the searches and the pose optimisation work on mvKeysUn
    return dict(n_f=len(kx), desc_f=np.ascontiguousarray(desc), kp_x=kx, kp_y=ky, kp_octave=np.ascontiguousarray(kps["octave"], np.int32),
                kp_angle=np.ascontiguousarray(kps["angle"], np.float32), u_right=ur.copy(), depth=dp.copy(),
                scale_factors=np.ascontiguousarray(scale_factors, np.float32), inv_sigma2=np.ascontiguousarray(inv_sigma2, np.float32),
                n_levels=len(scale_factors), min_x=np.float32(min_x), min_y=np.float32(min_y), max_x=np.float32(max_x), max_y=np.float32(max_y),
                grid_w_inv=gwi, grid_h_inv=ghi, grid_off=goff, grid_idx=gi, f_mp_state=np.zeros(max(len(kx), 1), np.uint8))


def frame_from_stereo(oe_l, oe_r, kps, desc, kps_r, desc_r, scen, scale_factors, inv_sigma2):
    """Frame::Frame(imLeft, imRight, ...) after the two ExtractORB calls (src/Frame.cc:57-113): the members of
    frame_from_extraction() with mvuRight / mvDepth from ComputeStereoMatches (:495-669; oe_l / oe_r hold the two pyramids)"""
    f = frame_from_extraction(kps, desc, np.zeros((scen["h"], scen["w"]), np.float32), scen, scale_factors, inv_sigma2)
    mbf = np.float32(scen["mbf"])
    mb = np.float32(mbf / np.float32(scen["fx"]))
    ur, dp, _ = O.compute_stereo_matches(oe_l, oe_r, kps, desc, kps_r, desc_r, mb, mbf)
    f["u_right"], f["depth"] = ur.copy(), dp.copy()
    return f


def _pose_problem(f, mp, table, Tcw, scen):
    idx = np.nonzero(mp >= 0)[0]
    rows = mp[idx]
    ur = f["u_right"][idx]
    return idx, dict(n=len(idx), Xw=table["pos"][rows].reshape(-1, 3), obs=np.stack([f["kp_x"][idx], f["kp_y"][idx], ur], 1).reshape(-1, 3),
                     stereo=(ur >= 0).astype(np.uint8), inv_sigma2=f["inv_sigma2"][f["kp_octave"][idx]],
                     fx=scen["fx"], fy=scen["fy"], cx=scen["cx"], cy=scen["cy"], bf=scen["mbf"], Tcw=np.asarray(Tcw, np.float32).reshape(16))


def _pose_optimization(f, mp, outlier, table, Tcw, scen):
    """Optimizer::PoseOptimization(Frame*): edges from the features that hold a map point (src/Optimizer.cc:260-350)"""
    idx, prob = _pose_problem(f, mp, table, Tcw, scen)
    outlier = outlier.copy()
    outlier[idx] = 0
    if prob["n"] == 0:
        return np.asarray(Tcw, np.float32).reshape(16).copy(), outlier, 0
    r = O.pose_optimization(prob)
    outlier[idx] = r["outlier"][: len(idx)]
    T = r["Tcw"].reshape(16) if prob["n"] >= 3 else np.asarray(Tcw, np.float32).reshape(16).copy()
    return T, outlier, r["n_inliers"]


def track_frame(f, last, table, local, Tcw_guess, Tlw, scen, th_last=15.0, th_local=3.0, nnratio_local=0.8, timing=None):
    """One CurrentFrame through the chain.  last: dict(mp, outlier, kp_octave, kp_angle) of the LastFrame; local: rows of
    the table (-1 = none).  Returns the Frame members after every stage."""
    def lap(name, t0):
        if timing is not None:
            timing[name] = timing.get(name, 0.0) + (time.perf_counter() - t0)
    out = {}
    fx, fy, cx, cy, mbf = (np.float32(scen[k]) for k in ("fx", "fy", "cx", "cy", "mbf"))
    n = f["n_f"]
    # ---- SearchByProjection(CurrentFrame, LastFrame, th, mSensor == MONOCULAR) (Tracking.cc:977-981)
    t0 = time.perf_counter()
    lmp = last["mp"]
    rows = np.where(lmp >= 0, lmp, 0)
    P = dict(n_last=len(lmp), last_valid=((lmp >= 0) & (last["outlier"] == 0)).astype(np.uint8), world_pos=table["pos"][rows],
             desc=table["desc"][rows], last_octave=np.ascontiguousarray(last["kp_octave"], np.int32),
             last_angle=np.ascontiguousarray(last["kp_angle"], np.float32), has_obs=(table["has_obs"][rows] * (lmp >= 0)).astype(np.uint8),
             Tcw=np.asarray(Tcw_guess, np.float32).reshape(16), Tlw=np.asarray(Tlw, np.float32).reshape(16), fx=fx, fy=fy, cx=cx, cy=cy,
             mb=np.float32(mbf / fx), mbf=mbf, th=np.float32(th_last), mono=0, check_orientation=1)
    fv = dict(f)
    fv["f_mp_state"] = np.zeros(max(n, 1), np.uint8)
    nm, match = O.search_by_projection_last(fv, P)
    mp = np.full(n, -1, np.int32)
    hit = match[:n] >= 0
    mp[hit] = lmp[match[:n][hit]]
    lap("search_by_projection_last", t0)
    out["nmatches_last"], out["mp_after_last"] = nm, mp.copy()
    # ---- PoseOptimization (:990) and the outlier discard (:1008-1025)
    t0 = time.perf_counter()
    outlier = np.zeros(n, np.uint8)
    T1, outlier, inl1 = _pose_optimization(f, mp, outlier, table, Tcw_guess, scen)
    lap("pose_optimization", t0)
    out["Tcw_1"], out["outlier_1"], out["inliers_1"] = T1.copy(), outlier.copy(), inl1
    seen = set(int(r) for r in mp[mp >= 0])
    disc = (mp >= 0) & (outlier != 0)
    mp[disc] = -1
    outlier[disc] = 0
    out["mp_after_discard"] = mp.copy()
    # ---- SearchLocalPoints (:1302-1352)
    t0 = time.perf_counter()
    T = T1.reshape(4, 4)
    R, t = np.ascontiguousarray(T[:3, :3]), np.ascontiguousarray(T[:3, 3])
    Rd, td = R.astype(np.float64), t.astype(np.float64)
    Ow = np.array([-((Rd[0, k] * td[0] + Rd[1, k] * td[1]) + Rd[2, k] * td[2]) for k in range(3)]).astype(np.float32)
    loc = np.asarray(local, np.int32)
    cand = np.array([r >= 0 and int(r) not in seen for r in loc], bool)
    rows = np.where(loc >= 0, loc, 0)
    pts = dict(n_pts=len(loc), valid=np.ones(len(loc), np.uint8), pos=table["pos"][rows], max_dist=table["max_dist"][rows],
               min_dist=table["min_dist"][rows], normal=table["normal"][rows], desc=table["desc"][rows],
               q_angle=np.zeros(len(loc), np.float32), R=R.reshape(9), t=t, Ow=Ow, R2=np.zeros(9, np.float32), t2=np.zeros(3, np.float32),
               fx=fx, fy=fy, cx=cx, cy=cy, bf=mbf, log_scale_factor=np.float32(np.log(np.float64(f["scale_factors"][1]))),
               inv_level_sigma2=f["inv_sigma2"], th=np.float32(th_local))
    fr = O.is_in_frustum(f, pts, 0.5)
    tiv = (fr["track_in_view"] * cand).astype(np.uint8)
    fv = dict(f)
    st = np.zeros(max(n, 1), np.uint8)
    st[:n] = np.where(mp >= 0, np.where(table["has_obs"][np.where(mp >= 0, mp, 0)] != 0, 2, 1), 0)
    fv["f_mp_state"] = st
    mpq = dict(n_mp=len(loc), track_in_view=tiv, pred_level=fr["pred_level"], view_cos=fr["view_cos"], proj_x=fr["proj_x"],
               proj_y=fr["proj_y"], proj_xr=fr["proj_xr"], desc=table["desc"][rows], has_obs=table["has_obs"][rows] * (loc >= 0),
               th=np.float32(th_local), nnratio=np.float32(nnratio_local))
    nm2, match2 = O.search_by_projection_mp(fv, mpq)
    hit = match2[:n] >= 0
    mp[hit] = loc[match2[:n][hit]]
    lap("search_local_points", t0)
    out["nmatches_local"], out["mp_after_local"], out["track_in_view"] = nm2, mp.copy(), tiv
    # ---- PoseOptimization (:1039)
    t0 = time.perf_counter()
    T2, outlier, inl2 = _pose_optimization(f, mp, outlier, table, T1, scen)
    lap("pose_optimization", t0)
    out["Tcw_2"], out["outlier_2"], out["inliers_2"] = T2.copy(), outlier.copy(), inl2
    return out

"""Synthetic tracking scenario for the per-frame chain (tests/, bench.py): B independent (LastFrame, CurrentFrame) pairs.

Datasets are absent, so a pair is a textured fronto-parallel plane seen by a translating RGB-D camera:
the current image is the last one shifted by an integer number of pixels plus pixel noise, the depth image is the
plane's depth with a few invalid holes, and a third ("older") view of the same plane provides local map points that
are not in the last frame.  Everything the chain needs besides the images -- the MapPoint table, LastFrame's
mvpMapPoints, the local map point lists, the motion-model pose guesses -- is derived once, from the keypoints the
extractor finds in the last / older views, by `build_map()`.  Pure numpy; deterministic per seed.
"""
from __future__ import annotations

import numpy as np

from . import synth


def _rot(axis, ang):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def pmap(fn, items, workers=None):
    """[fn(x) for x in items], on forked worker processes when there are enough items to pay for them (the generators are
    numpy + Python loops: 0.2 s per scene, 2 s per LocalBA window; a bench line holds hundreds).  The result does not depend on
    the number of workers (every item draws from a generator of its own).  Falls back to the serial loop."""
    import os
    items = list(items)
    if workers is None:
        workers = min(len(items) // 4, (os.cpu_count() or 1) - 2, 96)
    if int(os.environ.get("AOS2_SYNTH_WORKERS", workers)) <= 1 or len(items) < 8:
        return [fn(x) for x in items]
    workers = int(os.environ.get("AOS2_SYNTH_WORKERS", workers))
    try:
        import multiprocessing as mp
        # (forked from a process that may already hold the HIP runtime's threads: a child that inherits a held lock would hang, not
        # fail -- the children only run numpy, and the wait is bounded: past it the pool is dropped and the serial loop runs)
        with mp.get_context("fork").Pool(workers) as pool:
            job = pool.map_async(fn, items, chunksize=max(1, len(items) // (4 * workers)))
            return job.get(timeout=float(os.environ.get("AOS2_SYNTH_TIMEOUT", max(180.0, 6.0 * len(items)))))
    except Exception:   # noqa: BLE001  (no fork, no semaphores, a hung child, ...): the serial loop gives the same arrays
        return [fn(x) for x in items]


def _scene(a):
    """one distinct (LastFrame, CurrentFrame, older view) scene; every draw from the scene's own generator"""
    seed, u, W, H, M, max_shift, fx, fy, bf, stereo = a
    rng = np.random.default_rng([77000 + seed, u])
    DMAX = 64 if stereo else 0   # stereo: the canvas extends to the right by the largest disparity
    canvas8 = synth.synth_image(seed * 1000 + u, W + 2 * M + DMAX, H + 2 * M)
    canvas = canvas8.astype(np.float32)
    dx, dy = (int(v) for v in rng.integers(-max_shift, max_shift + 1, 2))
    ex, ey = (int(v) for v in rng.integers(-max_shift, max_shift + 1, 2))
    last = canvas[M:M + H, M:M + W].astype(np.uint8)
    cur = np.clip(np.rint(canvas[M + dy:M + dy + H, M + dx:M + dx + W] + rng.normal(0, 1.5, (H, W))), 0, 255).astype(np.uint8)
    old = np.clip(np.rint(canvas[M + ey:M + ey + H, M + ex:M + ex + W] + rng.normal(0, 1.0, (H, W))), 0, 255).astype(np.uint8)
    Z = rng.uniform(1.5, 3.0)
    right = None
    if stereo:
        # a rectified stereo camera (baseline bf / fx) in front of the plane: a feature at x in the left image appears at x - D in
        # the right one, D = bf / Z -- a whole number of pixels here (the plane's depth is chosen that way; the sub-pixel
        # refinement of ComputeStereoMatches then works on pixel noise only)
        D = int(rng.integers(6, DMAX - 3))
        Z = float(bf) / D
        right = np.clip(np.rint(canvas[M + dy:M + dy + H, M + dx + D:M + dx + D + W] + rng.normal(0, 1.5, (H, W))), 0, 255).astype(np.uint8)
    holes = np.zeros((2, 12, 4), np.int32)   # invalid-depth holes: monocular observations (mvuRight < 0)
    for d in range(2):
        for k in range(12):
            x0, y0 = int(rng.integers(0, W - 40)), int(rng.integers(0, H - 40))
            holes[d, k] = (y0, y0 + int(rng.integers(8, 40)), x0, x0 + int(rng.integers(8, 40)))
    # world frame: an arbitrary rigid transform of the last camera frame
    T = np.eye(4)
    T[:3, :3] = _rot(rng.normal(size=3), rng.uniform(0, np.pi))
    T[:3, 3] = rng.uniform(-5, 5, 3)
    # the camera translates parallel to the image plane: a feature at u_l in the last image appears at u_l - dx
    Tcl = np.eye(4)
    Tcl[:3, 3] = -np.array([dx * Z / float(fx), dy * Z / float(fy), 0.0])
    Tcw = Tcl @ T
    # motion-model guess (mVelocity * mLastFrame.mTcw): the true pose off by a fraction of a degree and ~2 px
    E = np.eye(4)
    E[:3, :3] = _rot(rng.normal(size=3), rng.normal(0, 0.002))
    E[:3, 3] = rng.normal(0, 2.0 * Z / float(fx), 3)
    return dict(canvas=canvas8, last=last, cur=cur, old=old, right=right, Z=Z, holes=holes, shift=(dx, dy), shift_old=(ex, ey), Tlw=T, Tcw=Tcw, Tguess=E @ Tcw)


def tracking_scenario(seed: int, batch: int, cfg: str = "tum", n_unique: int | None = None, max_shift: int = 10, dist=None, stereo: bool = False):
    """Images, depth maps and poses of `batch` (LastFrame, CurrentFrame) pairs (`n_unique` distinct ones, tiled).
    stereo: every CurrentFrame also has a right image (`right_cur`; the stereo Frame constructor, src/Frame.cc:57-113, takes
    mvuRight / mvDepth from ComputeStereoMatches instead of a depth image; the LastFrame / older views keep their depth maps:
    they are the scenario's given state)."""
    c = synth.CONFIGS[cfg]
    W, H = c["w"], c["h"]
    fx, fy, cx, cy, mbf = (np.float32(c[k]) for k in ("fx", "fy", "cx", "cy", "bf"))
    M = 2 * max_shift + 4
    nu = min(batch, n_unique or batch)
    sc = pmap(_scene, [(seed, u, W, H, M, max_shift, float(fx), float(fy), float(mbf), bool(stereo)) for u in range(nu)])
    last, cur, old = (np.stack([q[k] for q in sc]) for k in ("last", "cur", "old"))
    depth_cur = np.zeros((nu, H, W), np.float32)
    depth_last = np.zeros((nu, H, W), np.float32)
    for u, q in enumerate(sc):
        for d, dimg in enumerate((depth_cur[u], depth_last[u])):
            dimg[:] = np.float32(q["Z"])
            for y0, y1, x0, x1 in q["holes"][d]:
                dimg[y0:y1, x0:x1] = 0.0
    Z = np.array([q["Z"] for q in sc], np.float64)
    Tlw, Tcw, Tguess = (np.stack([q[k] for q in sc]) for k in ("Tlw", "Tcw", "Tguess"))
    shift = np.array([q["shift"] for q in sc], np.int32).reshape(nu, 2)
    shift_old = np.array([q["shift_old"] for q in sc], np.int32).reshape(nu, 2)
    idx = np.arange(batch) % nu
    # dist: mDistCoef (k1 k2 p1 p2 k3) the frames are declared to have (the images themselves are not warped: a parity scenario)
    return dict(seed=seed, margin=M, max_shift=max_shift, dist=None if dist is None else np.asarray(dist, np.float32), cfg=cfg, w=W, h=H, fx=fx, fy=fy, cx=cx, cy=cy, mbf=mbf, nfeatures=c["nfeatures"], batch=batch, n_unique=nu, index=idx,
                last=last, cur=cur, old=old, depth_cur=depth_cur, depth_last=depth_last, shift=shift, shift_old=shift_old, Z=Z,
                canvas=[q["canvas"] for q in sc], stereo=bool(stereo), right_cur=np.stack([q["right"] for q in sc]) if stereo else None,
                Tlw=Tlw.astype(np.float32), Tcw_true=Tcw.astype(np.float32), Tcw_guess=Tguess.astype(np.float32))


def build_map(scen: dict, last_kps, last_desc, old_kps, old_desc, scale_factors, n_local: int = 1500, seed: int = 0):
    """MapPoint table + LastFrame members + local map point lists from the keypoints of the last / older views.

    last_kps[u] / old_kps[u]: structured keypoint arrays (x, y, size, angle, response, octave, class_id) of unique pair u,
    last_desc[u] / old_desc[u]: their descriptors.  Returns numpy arrays with one slice per UNIQUE pair; rows of the
    table are global (pair u owns a contiguous range)."""
    nu = scen["n_unique"]
    fx, fy, cx, cy = (float(scen[k]) for k in ("fx", "fy", "cx", "cy"))
    rng = np.random.default_rng(99000 + seed)
    sf = np.asarray(scale_factors, np.float64)
    pos, desc, has_obs, normal, mind, maxd = [], [], [], [], [], []
    cap = max(max(len(k) for k in last_kps), 1)
    mp_last = np.full((nu, cap), -1, np.int32)
    outl_last = np.zeros((nu, cap), np.uint8)
    local = np.full((nu, n_local), -1, np.int32)
    row0 = 0
    for u in range(nu):
        Tlw = scen["Tlw"][u].astype(np.float64).reshape(4, 4)
        Rwl, twl = Tlw[:3, :3].T, -Tlw[:3, :3].T @ Tlw[:3, 3]
        Ow = twl   # centre of the last camera in world coordinates
        Z = scen["Z"][u]
        rows_last, rows_old = [], []

        def add(x_l, y_l, octave, d, obs, Z=Z):
            Xl = np.array([(x_l - cx) * Z / fx, (y_l - cy) * Z / fy, Z])
            Xw = Rwl @ Xl + twl
            n = Xw - Ow
            dist = np.linalg.norm(n)
            pos.append(Xw.astype(np.float32)); desc.append(d); has_obs.append(obs)
            normal.append((n / dist).astype(np.float32))
            mx = dist * sf[int(octave)]
            maxd.append(np.float32(mx)); mind.append(np.float32(np.float32(mx) / np.float32(sf[-1])))   # raw mfMaxDistance / mfMinDistance (src/MapPoint.cc:406-407)
            return len(pos) - 1

        dl = scen["depth_last"][u]
        real = bool(scen.get("real"))   # real frames: every map point takes the depth the sensor measured at its keypoint
        for i, k in enumerate(last_kps[u]):
            if dl[int(k["y"]), int(k["x"])] <= 0 or rng.random() < 0.08:
                continue   # no depth -> no map point was created for it; a few more stay NULL
            zk = float(dl[int(k["y"]), int(k["x"])]) if real else Z
            mp_last[u, i] = add(float(k["x"]), float(k["y"]), k["octave"], last_desc[u][i], 1 if rng.random() < 0.9 else 0, zk)
            rows_last.append(mp_last[u, i])
            if rng.random() < 0.03:
                outl_last[u, i] = 1   # LastFrame.mvbOutlier
        ex, ey = scen["shift_old"][u]
        for i, k in enumerate(old_kps[u]):
            if rng.random() < 0.5:
                continue
            zk = Z
            if real:   # (the "older" view of a real pair is the last frame itself: points the last frame does not hold)
                if mp_last[u, i] >= 0 or dl[int(k["y"]), int(k["x"])] <= 0:
                    continue
                zk = float(dl[int(k["y"]), int(k["x"])])
            rows_old.append(add(float(k["x"]) + ex, float(k["y"]) + ey, k["octave"], old_desc[u][i], 1, zk))
        junk = []
        for _ in range(60):   # points behind the camera / far outside the image: rejected by isInFrustum
            Xl = np.array([rng.uniform(-30, 30), rng.uniform(-30, 30), rng.uniform(-5, 0.5)])
            Xw = Rwl @ Xl + twl
            pos.append(Xw.astype(np.float32)); desc.append(rng.integers(0, 256, 32, dtype=np.uint8)); has_obs.append(1)
            normal.append(np.array([0, 0, 1], np.float32)); maxd.append(np.float32(100)); mind.append(np.float32(0.1))
            junk.append(len(pos) - 1)
        lst = (rows_old + rows_last + junk)[:n_local]
        local[u, :len(lst)] = lst
        row0 = len(pos)
    table = dict(n=len(pos), pos=np.stack(pos).astype(np.float32), desc=np.stack(desc).astype(np.uint8),
                 has_obs=np.asarray(has_obs, np.uint8), normal=np.stack(normal).astype(np.float32),
                 min_dist=np.asarray(mind, np.float32), max_dist=np.asarray(maxd, np.float32))
    return dict(table=table, mp_last=mp_last, outlier_last=outl_last, local=local, n_local=n_local)


def tracking_scenario_real(pairs: dict, batch: int, cfg: str = "tum"):
    """The scenario dict of tracking_scenario() from REAL (LastFrame, CurrentFrame) pairs (datasets.tum_pairs): images and
    depth maps as recorded; the world frame is the last camera's (mTcw of LastFrame = identity) and the motion-model
    guess for the current frame is "no motion" (consecutive frames of a 30 Hz sequence move by a few pixels: inside the
    15-pixel window of SearchByProjection(Current, Last)).  The map is still built from the extractor's keypoints by
    build_map(), every point at the depth the sensor measured there.  Ground-truth poses are not used (Tcw_true = the guess)."""
    c = synth.CONFIGS[cfg]
    nu = len(pairs["last"])
    H, W = pairs["last"].shape[1:]
    eye = np.tile(np.eye(4, dtype=np.float32), (nu, 1, 1))
    fx, fy, cx, cy, mbf = (np.float32(c[k]) for k in ("fx", "fy", "cx", "cy", "bf"))
    return dict(real=True, dist=None, cfg=cfg, w=W, h=H, fx=fx, fy=fy, cx=cx, cy=cy, mbf=mbf, nfeatures=c["nfeatures"], batch=batch,
                n_unique=nu, index=np.arange(batch) % nu, last=pairs["last"], cur=pairs["cur"], old=pairs["last"],
                depth_cur=pairs["depth_cur"], depth_last=pairs["depth_last"], shift=np.zeros((nu, 2), np.int32),
                shift_old=np.zeros((nu, 2), np.int32), Z=np.ones(nu), Tlw=eye.copy(), Tcw_true=eye.copy(), Tcw_guess=eye.copy())


def _neighbours(a):
    """the `n_nb` neighbour views of one scene; every draw from the scene's own generator"""
    seed, u, canvas8, W, H, M, max_shift, n_nb, fx, fy, cx, cy, T1w, Z = a
    rng = np.random.default_rng([55000 + seed, u])
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    Ki = np.linalg.inv(K)
    canvas = canvas8.astype(np.float32)
    imgs = np.zeros((n_nb, H, W), np.uint8)
    Tkw = np.zeros((n_nb, 4, 4), np.float32)
    F12 = np.zeros((n_nb, 9), np.float32)
    shift = np.zeros((n_nb, 2), np.int32)
    for k in range(n_nb):
        while True:
            dx, dy = (int(v) for v in rng.integers(-max_shift, max_shift + 1, 2))
            if dx or dy:
                break
        shift[k] = (dx, dy)
        imgs[k] = np.clip(np.rint(canvas[M + dy:M + dy + H, M + dx:M + dx + W] + rng.normal(0, 1.0, (H, W))), 0, 255).astype(np.uint8)
        T21 = np.eye(4)
        T21[:3, 3] = -np.array([dx * Z / fx, dy * Z / fy, 0.0])
        T2w = T21 @ T1w
        Tkw[k] = T2w
        R1w, t1w, R2w, t2w = T1w[:3, :3], T1w[:3, 3], T2w[:3, :3], T2w[:3, 3]
        R12 = R1w @ R2w.T
        t12 = -R12 @ t2w + t1w
        tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
        F12[k] = (Ki.T @ tx @ R12 @ Ki).astype(np.float32).reshape(9)
    return imgs, Tkw, F12, shift


def keyframe_neighbours(scen: dict, n_nb: int = 20, n_scenes: int | None = None):
    """Neighbour keyframes for the keyframe work of LocalMapping (CreateNewMapPoints / SearchInNeighbors, src/LocalMapping.cc:
    207-453, 455-560): `n_nb` more views of each of the first `n_scenes` distinct scenes of a tracking_scenario() (default: all),
    the camera translated parallel to the
    image plane like the scenario's other views (the image = the scene's canvas shifted by whole pixels + pixel noise).  Returns
    images [n_scenes * n_nb, H, W], their poses Tkw (float32 4x4) and, per (scene, neighbour), the fundamental matrix F12 between the
    scene's LastFrame view (keyframe 1) and the neighbour (keyframe 2) as LocalMapping::ComputeF12 forms it (:603-618):
    R12 = R1w R2w^T, t12 = -R12 t2w + t1w, F12 = K^-T [t12]x R12 K^-1."""
    if scen.get("real"):
        raise ValueError("neighbour keyframes are generated from the synthetic scenes' canvases")
    W, H, M, nu = scen["w"], scen["h"], scen["margin"], scen["n_unique"]
    ns = nu if n_scenes is None else min(int(n_scenes), nu)
    fx, fy, cx, cy = (float(scen[k]) for k in ("fx", "fy", "cx", "cy"))
    out = pmap(_neighbours, [(scen["seed"], u, scen["canvas"][u], W, H, M, scen["max_shift"], n_nb, fx, fy, cx, cy,
                              scen["Tlw"][u].astype(np.float64), float(scen["Z"][u])) for u in range(ns)])
    return dict(n_nb=n_nb, n_scenes=ns, imgs=np.concatenate([o[0] for o in out]), Tkw=np.concatenate([o[1] for o in out]),
                F12=np.concatenate([o[2] for o in out]), shift=np.concatenate([o[3] for o in out]))

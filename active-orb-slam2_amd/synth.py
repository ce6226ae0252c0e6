"""Seeded synthetic inputs for the ORB hot path (datasets are absent: SURVEY.md F7, §8(d)).

Pure numpy, deterministic per seed.  Used by tests/, bench.py and __graft_entry__.smoke().
"""
from __future__ import annotations

import numpy as np

# reference configs (Examples/RGB-D/TUM1.yaml, Examples/Stereo/KITTI00-02.yaml, Examples/Stereo/EuRoC.yaml)
CONFIGS = {
    "tum": dict(w=640, h=480, nfeatures=1000, fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, bf=40.0),
    "kitti": dict(w=1241, h=376, nfeatures=2000, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, bf=386.1448),
    "euroc": dict(w=752, h=480, nfeatures=1200, fx=435.2046959714599, fy=435.2046959714599, cx=367.4517211914062, cy=252.2008514404297, bf=47.90639384423901),
}
ORB_PARAMS = dict(scale_factor=1.2, nlevels=8, ini_th=20, min_th=7)


def _value_noise(rng, h, w, cell, amp):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.uniform(-amp, amp, size=(gh, gw)).astype(np.float32)
    ys = (np.arange(h, dtype=np.float32) / cell)
    xs = (np.arange(w, dtype=np.float32) / cell)
    y0 = ys.astype(np.int32)
    x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def synth_image(seed: int, w: int = 640, h: int = 480) -> np.ndarray:
    """u8 grayscale test image: 3-octave value noise + rectangles (+ a low-contrast zone that
    exercises the minThFAST fallback, src/ORBextractor.cc:812-816) + Gaussian pixel noise."""
    rng = np.random.default_rng(np.uint64(0x9E3779B97F4A7C15) ^ np.uint64(seed))
    img = np.full((h, w), 118.0, dtype=np.float32)
    img += _value_noise(rng, h, w, 64, 60)
    img += _value_noise(rng, h, w, 16, 40)
    img += _value_noise(rng, h, w, 4, 20)
    n_rect = 400
    cx = rng.integers(0, w, n_rect)
    cy = rng.integers(0, h, n_rect)
    sw = rng.integers(6, 61, n_rect)
    sh = rng.integers(6, 61, n_rect)
    val = rng.uniform(0, 255, n_rect).astype(np.float32)
    rot = rng.uniform(0, np.pi, n_rect)
    for i in range(n_rect):
        x0, x1 = max(0, cx[i] - sw[i] // 2), min(w, cx[i] + sw[i] // 2)
        y0, y1 = max(0, cy[i] - sh[i] // 2), min(h, cy[i] + sh[i] // 2)
        if x1 <= x0 or y1 <= y0:
            continue
        if i % 4 == 3:  # rotated rectangle
            R = int(0.75 * max(sw[i], sh[i])) + 1
            xa, xb = max(0, cx[i] - R), min(w, cx[i] + R)
            ya, yb = max(0, cy[i] - R), min(h, cy[i] + R)
            yy, xx = np.mgrid[ya:yb, xa:xb]
            c, s = np.cos(rot[i]), np.sin(rot[i])
            u = (xx - cx[i]) * c + (yy - cy[i]) * s
            v = -(xx - cx[i]) * s + (yy - cy[i]) * c
            m = (np.abs(u) <= sw[i] / 2) & (np.abs(v) <= sh[i] / 2)
            img[ya:yb, xa:xb][m] = val[i]
        else:
            img[y0:y1, x0:x1] = val[i]
    # low-contrast zone: smooth background + faint rectangles (contrast 9..18)
    zx0, zx1, zy0, zy1 = 0, w // 3, (2 * h) // 3, h
    zone = np.full((zy1 - zy0, zx1 - zx0), 100.0, dtype=np.float32)
    for _ in range(30):
        a = rng.integers(0, zx1 - zx0 - 8)
        b = rng.integers(0, zy1 - zy0 - 8)
        ww = rng.integers(6, 40)
        hh = rng.integers(6, 40)
        zone[b:b + hh, a:a + ww] += rng.uniform(9, 18) * rng.choice([-1.0, 1.0])
    img[zy0:zy1, zx0:zx1] = zone
    # one totally flat block (no corners at any threshold)
    img[0:h // 6, (5 * w) // 6:w] = 77.0
    noise = rng.normal(0.0, 3.0, size=(h, w)).astype(np.float32)
    noise[zy0:zy1, zx0:zx1] *= 0.3
    noise[0:h // 6, (5 * w) // 6:w] = 0.0
    img += noise
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synth_batch(seed0: int, n: int, w: int = 640, h: int = 480) -> np.ndarray:
    return np.stack([synth_image(seed0 + i, w, h) for i in range(n)], axis=0)


# ----------------------------------------------------------------------------- matching
def synth_stereo_pair(seed: int, w: int = 640, h: int = 480, max_disp: int = 48):
    """Rectified stereo pair: the right image is the left one resampled by a per-row-band fractional
    disparity (right(x) = left(x + d)), plus independent pixel noise.  Returns (left, right, disp_rows)."""
    wide = synth_image(seed, w + max_disp + 2, h).astype(np.float32)
    rng = np.random.default_rng(np.uint64(0xD1B54A32D192ED03) ^ np.uint64(seed))
    band = 32
    nb = (h + band - 1) // band
    dband = rng.uniform(3.0, max_disp - 1.0, nb).astype(np.float32)
    disp = np.repeat(dband, band)[:h]
    left = wide[:, :w]
    right = np.empty((h, w), np.float32)
    xs = np.arange(w)
    for y in range(h):
        d0 = int(np.floor(disp[y]))
        f = disp[y] - d0
        right[y] = (1 - f) * wide[y, xs + d0] + f * wide[y, xs + d0 + 1]
    right += rng.normal(0, 1.5, size=(h, w)).astype(np.float32)
    return (np.clip(np.rint(left), 0, 255).astype(np.uint8), np.clip(np.rint(right), 0, 255).astype(np.uint8),
            disp)


def synth_vocabulary(seed: int, k: int = 10, L: int = 3, stop_frac: float = 0.02, ragged: bool = False):
    """Hierarchical binary vocabulary in the reference's node order (DBoW2 saves nodes depth-first as k-means
    created them; here level by level -- both satisfy "parents precede children").  Each child descriptor is
    its parent's with bits flipped (fewer flips deeper), leaf weights are idf-like positives, a fraction of
    the words is stopped (weight 0, TemplatedVocabulary.h:1176).  ragged=True removes some subtrees so that
    leaves appear above level L.  Returns dict(k, L, scoring=0 (L1_NORM), weighting=0 (TF_IDF), parent, desc,
    weight, is_leaf) for nodes 1..n (node 0 is the root)."""
    rng = np.random.default_rng(np.uint64(0xA0761D6478BD642F) ^ np.uint64(seed))
    nid = 1
    prev_ids = np.array([0])
    prev_desc = rng.integers(0, 256, size=(1, 32), dtype=np.uint8)
    all_parent, all_desc, all_leaf = [], [], []
    for lev in range(1, L + 1):
        if ragged and lev > 1 and len(prev_ids) > 2:  # some nodes of the previous level stay childless (= leaves)
            keep = rng.random(len(prev_ids)) > 0.15
            keep[0] = True
            all_leaf[-1][~keep] = 1
            prev_ids, prev_desc = prev_ids[keep], prev_desc[keep]
        n_par = len(prev_ids)
        # bit flips: p = 1/2 below the root, then 1/8, 1/16, ... (AND of independent random bytes)
        flips = rng.integers(0, 256, size=(n_par, k, 32), dtype=np.uint8)
        for _ in range(0 if lev == 1 else min(lev, 4)):
            flips &= rng.integers(0, 256, size=(n_par, k, 32), dtype=np.uint8)
        d = prev_desc[:, None, :] ^ flips
        ids = nid + np.arange(n_par * k)
        all_parent.append(np.repeat(prev_ids, k).astype(np.int32))
        all_desc.append(d.reshape(-1, 32))
        all_leaf.append(np.full(n_par * k, 1 if lev == L else 0, np.uint8))
        prev_ids, prev_desc = ids, d.reshape(-1, 32)
        nid += n_par * k
    parent = np.concatenate(all_parent)
    desc = np.concatenate(all_desc)
    is_leaf = np.concatenate(all_leaf)
    weight = np.where(is_leaf > 0, rng.uniform(0.5, 12.0, len(parent)), 0.0)
    weight = weight.astype(np.float32).astype(np.float64)  # what the binary file format can hold
    stop = (rng.random(len(parent)) < stop_frac) & (is_leaf > 0)
    weight[stop] = 0.0
    return dict(k=k, L=L, scoring=0, weighting=0, parent=parent, desc=desc, weight=weight, is_leaf=is_leaf)


def vocab_descriptors(rng, voc, n, p=0.04):
    """n query descriptors: random leaves of the vocabulary with bit noise (plus 10 % unrelated ones)"""
    leaves = np.flatnonzero(voc["is_leaf"])
    pick = rng.choice(leaves, n)
    d = flip_bits(rng, voc["desc"][pick], p)
    r = rng.random(n) < 0.1
    d[r] = rng.integers(0, 256, size=(int(r.sum()), 32), dtype=np.uint8)
    return d


def synth_descriptors(rng, n):
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def flip_bits(rng, desc, p=0.08):
    """copy + flip each bit with probability p"""
    bits = np.unpackbits(desc, axis=1)
    flips = rng.random(bits.shape) < p
    return np.packbits(bits ^ flips.astype(np.uint8), axis=1)


def synth_bow_problem(seed: int, n_kf: int = 1000, n_f: int = 1000, n_nodes: int = 100,
                      nnratio: float = 0.7, check_orientation: bool = True):
    """Two descriptor sets with planted matches + FeatureVectors."""
    rng = np.random.default_rng(1000 + seed)
    desc_kf = synth_descriptors(rng, n_kf)
    desc_f = synth_descriptors(rng, n_f)
    node_kf = rng.integers(0, n_nodes, n_kf)
    node_f = rng.integers(0, n_nodes, n_f)
    angle_kf = rng.uniform(0, 360, n_kf).astype(np.float32)
    angle_f = rng.uniform(0, 360, n_f).astype(np.float32)
    n_pl = int(0.6 * min(n_kf, n_f))
    src = rng.permutation(n_kf)[:n_pl]
    dst = rng.permutation(n_f)[:n_pl]
    desc_f[dst] = flip_bits(rng, desc_kf[src], 0.08)
    same = rng.random(n_pl) < 0.9
    node_f[dst[same]] = node_kf[src[same]]
    dang = rng.normal(35.0, 5.0, n_pl).astype(np.float32)
    angle_f[dst] = np.mod(angle_kf[src] - dang, 360.0).astype(np.float32)
    kf_has_mp = (rng.random(n_kf) < 0.8).astype(np.uint8)

    def fv(nodes):
        order = np.argsort(nodes, kind="stable")
        ids, counts = np.unique(nodes[order], return_counts=True)
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        return ids.astype(np.int32), off, order.astype(np.int32)

    id_kf, off_kf, idx_kf = fv(node_kf)
    id_f, off_f, idx_f = fv(node_f)
    return dict(desc_kf=desc_kf, desc_f=desc_f, kf_has_mp=kf_has_mp, angle_kf=angle_kf,
                angle_f=angle_f, node_id_kf=id_kf, node_off_kf=off_kf, node_idx_kf=idx_kf,
                node_id_f=id_f, node_off_f=off_f, node_idx_f=idx_f, nnratio=np.float32(nnratio),
                check_orientation=int(check_orientation))


GRID_COLS, GRID_ROWS = 64, 48  # include/Frame.h:37-38


def _feature_vector(nodes):
    order = np.argsort(nodes, kind="stable")
    ids, counts = np.unique(nodes[order], return_counts=True)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return ids.astype(np.int32), off, order.astype(np.int32)


def synth_bow_kf_problem(seed: int, n1: int = 1000, n2: int = 1000, n_nodes: int = 100, nnratio: float = 0.75,
                         check_orientation: bool = True):
    """Two keyframes with map points for SearchByBoW(KF, KF) (src/ORBmatcher.cc:522-655)."""
    b = synth_bow_problem(seed + 7000, n1, n2, n_nodes, nnratio, check_orientation)
    rng = np.random.default_rng(7100 + seed)
    return dict(desc1=b["desc_kf"], desc2=b["desc_f"], has_mp1=b["kf_has_mp"],
                has_mp2=(rng.random(n2) < 0.8).astype(np.uint8), angle1=b["angle_kf"], angle2=b["angle_f"],
                node_id1=b["node_id_kf"], node_off1=b["node_off_kf"], node_idx1=b["node_idx_kf"],
                node_id2=b["node_id_f"], node_off2=b["node_off_f"], node_idx2=b["node_idx_f"],
                nnratio=np.float32(nnratio), check_orientation=int(check_orientation))


def synth_triang_problem(seed: int, n1: int = 1000, n2: int = 1000, n_nodes: int = 100, cfg: str = "kitti",
                         only_stereo: bool = False, check_orientation: bool = True, mono: bool = False):
    """Two keyframes of a moving camera for SearchForTriangulation (src/ORBmatcher.cc:657-823): 60 % of the
    features are projections of common 3-D points (epipolar-consistent, similar descriptors, same vocabulary
    node), the rest is clutter.  F12 = K^-T [t12]x R12 K^-1 as LocalMapping::ComputeF12 builds it."""
    rng = np.random.default_rng(8000 + seed)
    c = CONFIGS[cfg]
    fx, fy, cx, cy, W, H = c["fx"], c["fy"], c["cx"], c["cy"], c["w"], c["h"]
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    ang = 0.03
    R2w = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t2w = np.array([-0.6, 0.02, 0.15])                    # camera 1 = world
    n_pl = int(0.6 * min(n1, n2))
    X = np.stack([rng.uniform(-8, 8, n_pl), rng.uniform(-3, 3, n_pl), rng.uniform(4, 30, n_pl)], 1)
    p1 = (K @ X.T).T
    p1 = p1[:, :2] / p1[:, 2:]
    Xc2 = (R2w @ X.T).T + t2w
    p2 = (K @ Xc2.T).T
    p2 = p2[:, :2] / p2[:, 2:]
    x1 = rng.uniform(0, W, n1).astype(np.float32)
    y1 = rng.uniform(0, H, n1).astype(np.float32)
    x2 = rng.uniform(0, W, n2).astype(np.float32)
    y2 = rng.uniform(0, H, n2).astype(np.float32)
    src, dst = rng.permutation(n1)[:n_pl], rng.permutation(n2)[:n_pl]
    x1[src], y1[src] = p1[:, 0], p1[:, 1]
    x2[dst], y2[dst] = p2[:, 0] + rng.normal(0, 0.4, n_pl), p2[:, 1] + rng.normal(0, 0.4, n_pl)
    desc1, desc2 = synth_descriptors(rng, n1), synth_descriptors(rng, n2)
    desc2[dst] = flip_bits(rng, desc1[src], 0.06)
    extra = rng.permutation(n2)[: n2 // 10]               # decoys: similar descriptor, wrong place
    desc2[extra] = flip_bits(rng, desc1[rng.integers(0, n1, len(extra))], 0.06)
    node1, node2 = rng.integers(0, n_nodes, n1), rng.integers(0, n_nodes, n2)
    node2[dst] = node1[src]
    angle1 = rng.uniform(0, 360, n1).astype(np.float32)
    angle2 = rng.uniform(0, 360, n2).astype(np.float32)
    angle2[dst] = np.mod(angle1[src] - rng.normal(10.0, 4.0, n_pl), 360.0).astype(np.float32)
    octave2 = rng.integers(0, 8, n2).astype(np.int32)
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    ur1 = np.where(rng.random(n1) < (0.0 if mono else 0.6), x1 - 3.0, -1.0).astype(np.float32)
    ur2 = np.where(rng.random(n2) < (0.0 if mono else 0.6), x2 - 3.0, -1.0).astype(np.float32)
    # R12, t12 (camera 2 -> camera 1): R12 = R1w R2w^T, t12 = -R12 t2w (+ t1w = 0)
    R12 = R2w.T
    t12 = -R12 @ t2w
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    F12 = np.linalg.inv(K).T @ tx @ R12 @ np.linalg.inv(K)
    C2 = t2w                                              # camera-1 centre (0) in camera 2
    ex, ey = fx * C2[0] / C2[2] + cx, fy * C2[1] / C2[2] + cy
    id1, off1, idx1 = _feature_vector(node1)
    id2, off2, idx2 = _feature_vector(node2)
    return dict(desc1=desc1, desc2=desc2, has_mp1=(rng.random(n1) < 0.4).astype(np.uint8),
                has_mp2=(rng.random(n2) < 0.4).astype(np.uint8), x1=x1, y1=y1, angle1=angle1, u_right1=ur1,
                x2=x2, y2=y2, angle2=angle2, u_right2=ur2, octave2=octave2, scale_factors2=sf,
                level_sigma2_2=(sf * sf).astype(np.float32), F12=F12.astype(np.float32).reshape(9),
                ex=np.float32(ex), ey=np.float32(ey), only_stereo=int(only_stereo),
                check_orientation=int(check_orientation), node_id1=id1, node_off1=off1, node_idx1=idx1,
                node_id2=id2, node_off2=off2, node_idx2=idx2)


def synth_observations(seed: int, n_points: int = 2000, max_obs: int = 24):
    """Observed descriptors of a batch of map points (CSR) for MapPoint::ComputeDistinctiveDescriptors."""
    rng = np.random.default_rng(9000 + seed)
    cnt = rng.integers(0, max_obs + 1, n_points)
    cnt[rng.random(n_points) < 0.02] = max_obs * 4        # a few heavily observed points
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    base = synth_descriptors(rng, n_points)
    desc = np.repeat(base, cnt, axis=0)
    desc = flip_bits(rng, desc, 0.08) if len(desc) else desc
    return off, desc


def build_grid(kp_x, kp_y, min_x, min_y, max_x, max_y):
    """Frame::AssignFeaturesToGrid (src/Frame.cc:259-274) as CSR, cell = ix*48+iy."""
    gw_inv = np.float32(GRID_COLS) / np.float32(max_x - min_x)
    gh_inv = np.float32(GRID_ROWS) / np.float32(max_y - min_y)
    px = np.round((kp_x - np.float32(min_x)) * gw_inv).astype(np.int64)   # round half away (x>=0)
    py = np.round((kp_y - np.float32(min_y)) * gh_inv).astype(np.int64)
    # C round() is half-away-from-zero; numpy rounds half to even -> redo exactly
    fxv = (kp_x - np.float32(min_x)) * gw_inv
    fyv = (kp_y - np.float32(min_y)) * gh_inv
    px = np.where(fxv >= 0, np.floor(fxv + np.float32(0.5)), np.ceil(fxv - np.float32(0.5))).astype(np.int64)
    py = np.where(fyv >= 0, np.floor(fyv + np.float32(0.5)), np.ceil(fyv - np.float32(0.5))).astype(np.int64)
    ok = (px >= 0) & (px < GRID_COLS) & (py >= 0) & (py < GRID_ROWS)
    cell = px * GRID_ROWS + py
    idx = np.nonzero(ok)[0]
    order = np.argsort(cell[idx], kind="stable")
    idx = idx[order]
    counts = np.bincount(cell[idx], minlength=GRID_COLS * GRID_ROWS)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return off, idx.astype(np.int32), np.float32(gw_inv), np.float32(gh_inv)


def synth_frame_view(rng, n_f, w, h, n_levels=8, stereo=True, frac_with_obs=0.1):
    kp_x = rng.uniform(20, w - 20, n_f).astype(np.float32)
    kp_y = rng.uniform(20, h - 20, n_f).astype(np.float32)
    octave = np.minimum(rng.geometric(0.35, n_f) - 1, n_levels - 1).astype(np.int32)
    angle = rng.uniform(0, 360, n_f).astype(np.float32)
    u_right = np.where(rng.random(n_f) < (0.7 if stereo else 0.0), kp_x - rng.uniform(2, 40, n_f), -1.0).astype(np.float32)
    sf = (np.float32(1.2) ** np.arange(n_levels)).astype(np.float32)
    sf = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(n_levels - 1, np.float32(1.2), np.float32)])).astype(np.float32)
    min_x, min_y, max_x, max_y = np.float32(0), np.float32(0), np.float32(w), np.float32(h)
    off, idx, gwi, ghi = build_grid(kp_x, kp_y, min_x, min_y, max_x, max_y)
    state = rng.choice([0, 1, 2], size=n_f, p=[1 - frac_with_obs - 0.05, 0.05, frac_with_obs]).astype(np.uint8)
    desc = synth_descriptors(rng, n_f)
    return dict(n_f=n_f, desc_f=desc, kp_x=kp_x, kp_y=kp_y, kp_octave=octave, kp_angle=angle,
                u_right=u_right, scale_factors=sf, n_levels=n_levels, min_x=min_x, min_y=min_y,
                max_x=max_x, max_y=max_y, grid_w_inv=gwi, grid_h_inv=ghi, grid_off=off,
                grid_idx=idx, f_mp_state=state)


def _view_from(rng, kp_x, kp_y, w, h, n_levels=8, stereo=True, frac_with_mp=0.15):
    """frame / keyframe view around given keypoint positions (grid built from them)"""
    n_f = len(kp_x)
    kp_x, kp_y = kp_x.astype(np.float32), kp_y.astype(np.float32)
    octave = np.minimum(rng.geometric(0.35, n_f) - 1, n_levels - 1).astype(np.int32)
    angle = rng.uniform(0, 360, n_f).astype(np.float32)
    u_right = np.where(rng.random(n_f) < (0.7 if stereo else 0.0), kp_x - rng.uniform(2, 40, n_f), -1.0).astype(np.float32)
    sf = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(n_levels - 1, np.float32(1.2), np.float32)])).astype(np.float32)
    min_x, min_y, max_x, max_y = np.float32(0), np.float32(0), np.float32(w), np.float32(h)
    off, idx, gwi, ghi = build_grid(kp_x, kp_y, min_x, min_y, max_x, max_y)
    state = (rng.random(n_f) < frac_with_mp).astype(np.uint8)
    return dict(n_f=n_f, desc_f=synth_descriptors(rng, n_f), kp_x=kp_x, kp_y=kp_y, kp_octave=octave, kp_angle=angle,
                u_right=u_right, scale_factors=sf, n_levels=n_levels, min_x=min_x, min_y=min_y, max_x=max_x, max_y=max_y,
                grid_w_inv=gwi, grid_h_inv=ghi, grid_off=off, grid_idx=idx, f_mp_state=state)


def _small_pose(rv, t):
    th_ = np.linalg.norm(rv)
    K = np.array([[0, -rv[2], rv[1]], [rv[2], 0, -rv[0]], [-rv[1], rv[0], 0]])
    R = np.eye(3) + np.sin(th_) / th_ * K + (1 - np.cos(th_)) / th_ ** 2 * K @ K
    return R, np.asarray(t, np.float64)


def _points_for_view(rng, view, tgt, R, t, c, jitter=1.5):
    """world points whose projection with (R, t) lands near the view's keypoints `tgt`, with scale-invariance
    distances that predict the keypoint's octave (or one above), normals roughly facing the camera"""
    n = len(tgt)
    fx, fy, cx, cy = c["fx"], c["fy"], c["cx"], c["cy"]
    z = rng.uniform(1.0, 15.0, n)
    u = view["kp_x"][tgt] + rng.normal(0, jitter, n)
    v = view["kp_y"][tgt] + rng.normal(0, jitter, n)
    Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], axis=1)
    Xw = (Xc - t) @ R
    Ow = -R.T @ t
    PO = Xw - Ow
    dist = np.linalg.norm(PO, axis=1)
    lvl = view["kp_octave"][tgt] + rng.uniform(-0.95, 0.95, n)
    max_dist = dist * 1.2 ** lvl
    min_dist = max_dist / 1.2 ** 7                         # raw mfMinDistance / mfMaxDistance (the searches apply 0.8f / 1.2f)
    far = rng.random(n) < 0.05
    max_dist[far] *= 0.2                                   # out of the scale-invariance range
    nrm = PO / dist[:, None] + rng.normal(0, 0.25, (n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    back = rng.random(n) < 0.08
    nrm[back] *= -1                                        # viewed from behind (> 60 deg)
    behind = rng.random(n) < 0.03
    Xw[behind] = Ow + (Ow - Xw[behind])                    # negative depth
    return Xw, max_dist, min_dist, nrm, Ow


def synth_proj_gen_problem(seed: int, n_f: int = 1000, n_pts: int = 1500, cfg: str = "kitti", th: float = 3.0,
                           stereo: bool = True):
    """A keyframe / frame view + map points to project into it: the inputs of Fuse (src/ORBmatcher.cc:825-975,
    :977-1100), SearchByProjection(KF, Scw) (:290-403) and SearchByProjection(F, KF) (:1472-1599)."""
    rng = np.random.default_rng(11000 + seed)
    c = CONFIGS[cfg]
    w, h = c["w"], c["h"]
    view = _view_from(rng, rng.uniform(20, w - 20, n_f), rng.uniform(20, h - 20, n_f), w, h, stereo=stereo)
    R, t = _small_pose(np.array([0.02, -0.03, 0.01]), [0.3, -0.1, 0.2])
    tgt = rng.integers(0, n_f, n_pts)
    Xw, max_dist, min_dist, nrm, Ow = _points_for_view(rng, view, tgt, R, t, c)
    desc = flip_bits(rng, view["desc_f"][tgt], 0.08)
    rnd = rng.random(n_pts) < 0.2
    desc[rnd] = synth_descriptors(rng, int(rnd.sum()))
    q_angle = np.mod(view["kp_angle"][tgt] + rng.normal(15, 5, n_pts), 360).astype(np.float32)
    odd = rng.random(n_pts) < 0.15
    q_angle[odd] = rng.uniform(0, 360, int(odd.sum())).astype(np.float32)
    sf = view["scale_factors"]
    p = dict(n_pts=n_pts, valid=(rng.random(n_pts) < 0.85).astype(np.uint8), pos=Xw.astype(np.float32),
             max_dist=max_dist.astype(np.float32), min_dist=min_dist.astype(np.float32), normal=nrm.astype(np.float32),
             desc=desc, q_angle=q_angle, R=R.astype(np.float32).reshape(9), t=t.astype(np.float32),
             Ow=Ow.astype(np.float32), R2=np.eye(3, dtype=np.float32).reshape(9), t2=np.zeros(3, np.float32),
             fx=np.float32(c["fx"]), fy=np.float32(c["fy"]), cx=np.float32(c["cx"]), cy=np.float32(c["cy"]),
             bf=np.float32(c["bf"]), log_scale_factor=np.float32(np.log(np.float32(1.2))),
             inv_level_sigma2=(1.0 / (sf * sf)).astype(np.float32), th=np.float32(th))
    return view, p



def frustum_hand_case():
    """Hand-computed Frame::isInFrustum / MapPoint::PredictScale cases (src/Frame.cc:326-343, src/MapPoint.cc:413-459): a
    camera at the origin looking down +z, points on the optical axis at distance d with raw mfMaxDistance M and
    mfMinDistance m.  The range gate is 0.8f*m <= d <= 1.2f*M, the level is ceil(log(M / d) / log(1.2)) clamped to 0..7.
    Returns (frame view, point set, expected in_view, expected level)."""
    c = CONFIGS["tum"]
    rng = np.random.default_rng(5)
    view = _view_from(rng, rng.uniform(20, c["w"] - 20, 50), rng.uniform(20, c["h"] - 20, 50), c["w"], c["h"], stereo=True)
    #        d     M                 m     in_view level
    rows = [(5.0, 5.0 * 1.2 ** 2.5, 0.5, 1, 3),     # ratio 1.2^2.5 -> ceil(2.5) = 3 (the 1.2x-conflated value would give 4)
            (5.0, 5.0 * 1.2 ** 0.5, 0.5, 1, 1),     # ratio 1.2^0.5 -> 1
            (5.0, 5.0 / 1.1, 0.5, 1, 0),            # beyond the raw maximum but inside 1.2 * M: in view, ratio < 1 -> level 0
            (5.0, 5.0 / 1.25, 0.5, 0, 0),           # beyond 1.2 * M: rejected
            (5.0, 40.0, 5.0 / 0.85, 1, 7),          # below the raw minimum but above 0.8 * m: in view; level clamps to 7
            (5.0, 40.0, 5.0 / 0.75, 0, 0),          # below 0.8 * m: rejected
            (2.0, 2.0 * 1.2 ** 6.5, 0.1, 1, 7),     # ceil(6.5) = 7
            (2.0, 2.0 * 1.2 ** 5.5, 0.1, 1, 6)]
    n = len(rows)
    pos = np.array([[0, 0, r[0]] for r in rows], np.float32)
    p = dict(n_pts=n, valid=np.ones(n, np.uint8), pos=pos, max_dist=np.array([r[1] for r in rows], np.float32),
             min_dist=np.array([r[2] for r in rows], np.float32), normal=np.tile(np.array([0, 0, 1], np.float32), (n, 1)),
             desc=synth_descriptors(rng, n), q_angle=np.zeros(n, np.float32), R=np.eye(3, dtype=np.float32).reshape(9),
             t=np.zeros(3, np.float32), Ow=np.zeros(3, np.float32), R2=np.eye(3, dtype=np.float32).reshape(9), t2=np.zeros(3, np.float32),
             fx=np.float32(c["fx"]), fy=np.float32(c["fy"]), cx=np.float32(c["cx"]), cy=np.float32(c["cy"]), bf=np.float32(c["bf"]),
             log_scale_factor=np.float32(np.log(np.float32(1.2))), inv_level_sigma2=(1.0 / (view["scale_factors"] ** 2)).astype(np.float32),
             th=np.float32(3.0))
    return view, p, np.array([r[3] for r in rows], np.uint8), np.array([r[4] for r in rows], np.int32)


def synth_sim3_problem(seed: int, n1: int = 1000, n2: int = 1000, cfg: str = "kitti", th: float = 7.5):
    """Two keyframes related by a similarity for SearchBySim3 (src/ORBmatcher.cc:1102-1326): 50 % of the
    features observe common 3-D points (so the two projections can agree), the rest is clutter."""
    rng = np.random.default_rng(12000 + seed)
    c = CONFIGS[cfg]
    w, h, fx, fy, cx, cy = c["w"], c["h"], c["fx"], c["fy"], c["cx"], c["cy"]
    R1w, t1w = _small_pose(np.array([0.01, 0.02, -0.01]), [0.1, 0.0, 0.05])
    R2w, t2w = _small_pose(np.array([-0.02, 0.05, 0.01]), [-0.5, 0.05, 0.1])
    s12 = 1.07
    R12 = R1w @ R2w.T                                       # camera 2 -> camera 1 (up to the scale drift s12)
    t12 = t1w - s12 * R12 @ t2w
    sR12, sR21 = s12 * R12, (1.0 / s12) * R12.T
    t21 = -sR21 @ t12
    k = int(0.5 * min(n1, n2))
    x1, y1 = rng.uniform(20, w - 20, n1), rng.uniform(20, h - 20, n1)
    x2, y2 = rng.uniform(20, w - 20, n2), rng.uniform(20, h - 20, n2)
    i1, i2 = rng.permutation(n1)[:k], rng.permutation(n2)[:k]
    z = rng.uniform(3.0, 20.0, k)
    c1 = np.stack([rng.uniform(-0.7, 0.7, k) * z, rng.uniform(-0.25, 0.25, k) * z, z], 1)   # camera-1 coordinates
    c2 = (sR21 @ c1.T).T + t21
    x1[i1], y1[i1] = fx * c1[:, 0] / c1[:, 2] + cx, fy * c1[:, 1] / c1[:, 2] + cy
    x2[i2], y2[i2] = fx * c2[:, 0] / c2[:, 2] + cx, fy * c2[:, 1] / c2[:, 2] + cy
    ok = (x1[i1] > 5) & (x1[i1] < w - 5) & (y1[i1] > 5) & (y1[i1] < h - 5) & (x2[i2] > 5) & (x2[i2] < w - 5) & \
         (y2[i2] > 5) & (y2[i2] < h - 5) & (c2[:, 2] > 0.5)
    x1[i1[~ok]], y1[i1[~ok]] = rng.uniform(20, w - 20, int((~ok).sum())), rng.uniform(20, h - 20, int((~ok).sum()))
    x2[i2[~ok]], y2[i2[~ok]] = rng.uniform(20, w - 20, int((~ok).sum())), rng.uniform(20, h - 20, int((~ok).sum()))
    f1, f2 = _view_from(rng, x1, y1, w, h), _view_from(rng, x2, y2, w, h)
    f2["kp_octave"][i2] = f1["kp_octave"][i1]
    lsf = np.float32(np.log(np.float32(1.2)))

    def side(n, view, other_view, Rw, tw, R2_, t2_, mine, theirs, cam):
        # map points of this keyframe: world position = back-projection of its own features
        zz = rng.uniform(3.0, 20.0, n)
        cc = np.stack([(view["kp_x"] - cx) / fx * zz, (view["kp_y"] - cy) / fy * zz, zz], 1)
        cc[mine[ok]] = cam[ok]
        Xw = (cc - tw) @ Rw
        tgt_c = (R2_ @ cc.T).T + t2_                        # in the other camera
        dist = np.linalg.norm(tgt_c, axis=1)
        lvl = view["kp_octave"] + rng.uniform(-0.9, 0.9, n)
        max_dist = dist * 1.2 ** lvl
        desc = synth_descriptors(rng, n)
        desc[mine[ok]] = flip_bits(rng, other_view["desc_f"][theirs[ok]], 0.07)
        return dict(n_pts=n, valid=(rng.random(n) < 0.9).astype(np.uint8), pos=Xw.astype(np.float32),
                    max_dist=max_dist.astype(np.float32), min_dist=(max_dist / 1.2 ** 7).astype(np.float32),
                    normal=np.zeros((n, 3), np.float32), desc=desc, q_angle=np.zeros(n, np.float32),
                    R=Rw.astype(np.float32).reshape(9), t=tw.astype(np.float32), Ow=np.zeros(3, np.float32),
                    R2=R2_.astype(np.float32).reshape(9), t2=t2_.astype(np.float32), fx=np.float32(fx), fy=np.float32(fy),
                    cx=np.float32(cx), cy=np.float32(cy), bf=np.float32(c["bf"]), log_scale_factor=lsf,
                    inv_level_sigma2=(1.0 / (view["scale_factors"] ** 2)).astype(np.float32), th=np.float32(th))

    p12 = side(n1, f1, f2, R1w, t1w, sR21, t21, i1, i2, c1)
    p21 = side(n2, f2, f1, R2w, t2w, sR12, t12, i2, i1, c2)
    return f1, f2, p12, p21


def synth_init_problem(seed: int, n1: int = 1500, n2: int = 1500, w: int = 640, h: int = 480):
    """Two monocular frames a small motion apart for SearchForInitialization (src/ORBmatcher.cc:405-520):
    60 % of F1's features reappear in F2 displaced by a smooth flow; several F1 features compete for the
    same F2 feature (the vMatchedDistance / vnMatches21 stealing logic)."""
    rng = np.random.default_rng(13000 + seed)
    x1, y1 = rng.uniform(20, w - 20, n1), rng.uniform(20, h - 20, n1)
    x2, y2 = rng.uniform(20, w - 20, n2), rng.uniform(20, h - 20, n2)
    k = int(0.6 * min(n1, n2))
    i1, i2 = rng.permutation(n1)[:k], rng.permutation(n2)[:k]
    x2[i2] = np.clip(x1[i1] + 12 + 0.02 * (y1[i1] - h / 2) + rng.normal(0, 1.0, k), 1, w - 1)
    y2[i2] = np.clip(y1[i1] - 5 + rng.normal(0, 1.0, k), 1, h - 1)
    f2 = _view_from(rng, x2, y2, w, h, stereo=False)
    octave1 = np.minimum(rng.geometric(0.5, n1) - 1, 7).astype(np.int32)
    f2["kp_octave"][i2] = octave1[i1]
    desc1 = synth_descriptors(rng, n1)
    desc1[i1] = flip_bits(rng, f2["desc_f"][i2], 0.06)
    rivals = rng.permutation(n1)[: n1 // 8]                 # near-duplicates that fight for the same F2 feature
    src = rng.choice(i1, len(rivals))
    desc1[rivals] = flip_bits(rng, desc1[src], 0.03)
    x1[rivals], y1[rivals] = x1[src] + rng.normal(0, 3, len(rivals)), y1[src] + rng.normal(0, 3, len(rivals))
    octave1[rivals] = octave1[src]
    angle1 = rng.uniform(0, 360, n1).astype(np.float32)
    angle1[i1] = np.mod(f2["kp_angle"][i2] + rng.normal(8, 4, k), 360).astype(np.float32)
    prev = np.stack([x1, y1], 1).astype(np.float32)         # vbPrevMatched starts as F1's own keypoints
    return f2, dict(desc1=desc1, octave1=octave1, angle1=angle1, prev_xy=prev)


def synth_proj_mp_problem(seed: int, n_f: int = 1000, n_mp: int = 1500, w: int = 640, h: int = 480,
                          th: float = 3.0, nnratio: float = 0.8):
    """Frame + local map points with planted projections (SearchByProjection(F, vpMP, th))."""
    rng = np.random.default_rng(2000 + seed)
    f = synth_frame_view(rng, n_f, w, h)
    tgt = rng.integers(0, n_f, n_mp)
    proj_x = (f["kp_x"][tgt] + rng.normal(0, 1.5, n_mp)).astype(np.float32)
    proj_y = (f["kp_y"][tgt] + rng.normal(0, 1.5, n_mp)).astype(np.float32)
    ur = f["u_right"][tgt]
    proj_xr = np.where(ur > 0, ur + rng.normal(0, 1.0, n_mp), proj_x - 10).astype(np.float32)
    pred_level = np.clip(f["kp_octave"][tgt] + rng.integers(0, 2, n_mp), 0, 7).astype(np.int32)
    desc = flip_bits(rng, f["desc_f"][tgt], 0.1)
    rnd = rng.random(n_mp) < 0.25
    desc[rnd] = synth_descriptors(rng, int(rnd.sum()))
    mp = dict(n_mp=n_mp, track_in_view=(rng.random(n_mp) < 0.85).astype(np.uint8),
              pred_level=pred_level, view_cos=rng.uniform(0.9, 1.0, n_mp).astype(np.float32),
              proj_x=proj_x, proj_y=proj_y, proj_xr=proj_xr, desc=desc,
              has_obs=(rng.random(n_mp) < 0.9).astype(np.uint8), th=np.float32(th),
              nnratio=np.float32(nnratio))
    return f, mp


def synth_proj_last_problem(seed: int, n: int = 1000, w: int = 640, h: int = 480, cfg: str = "tum",
                            th: float = 7.0, mono: bool = False, check_orientation: bool = True):
    """Current frame + last frame map points (SearchByProjection(Cur, Last, th, bMono))."""
    rng = np.random.default_rng(3000 + seed)
    c = CONFIGS[cfg]
    cur = synth_frame_view(rng, n, w, h)
    fx, fy, cx, cy, bf = (np.float32(c[k]) for k in ("fx", "fy", "cx", "cy", "bf"))
    # current pose: small rotation + translation; last pose: identity-ish
    def pose(rv, t):
        th_ = np.linalg.norm(rv)
        K = np.array([[0, -rv[2], rv[1]], [rv[2], 0, -rv[0]], [-rv[1], rv[0], 0]])
        R = np.eye(3) + np.sin(th_) / th_ * K + (1 - np.cos(th_)) / th_ ** 2 * K @ K
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = t
        return T.astype(np.float32)
    fwd = [0.0, 0.0, -0.3, 0.3, 0.02][seed % 5]
    Tcw = pose(np.array([0.01, -0.02, 0.015]), np.array([0.05, -0.02, fwd]))
    Tlw = pose(np.array([0.001, 0.002, -0.001]), np.array([0.0, 0.0, 0.0]))
    # world points that project onto current keypoints (+ jitter)
    tgt = rng.integers(0, n, n)
    z = rng.uniform(1.0, 12.0, n)
    u = cur["kp_x"][tgt] + rng.normal(0, 2.0, n)
    v = cur["kp_y"][tgt] + rng.normal(0, 2.0, n)
    Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], axis=1)
    R, t = Tcw[:3, :3].astype(np.float64), Tcw[:3, 3].astype(np.float64)
    Xw = (Xc - t) @ R  # R^T (Xc - t)
    behind = rng.random(n) < 0.03
    Xw[behind] *= -1
    desc = flip_bits(rng, cur["desc_f"][tgt], 0.1)
    rnd = rng.random(n) < 0.2
    desc[rnd] = synth_descriptors(rng, int(rnd.sum()))
    last_angle = np.mod(cur["kp_angle"][tgt] + rng.normal(20, 6, n), 360).astype(np.float32)
    p = dict(n_last=n, last_valid=(rng.random(n) < 0.8).astype(np.uint8),
             world_pos=Xw.astype(np.float32), desc=desc,
             last_octave=np.clip(cur["kp_octave"][tgt] + rng.integers(-1, 2, n), 0, 7).astype(np.int32),
             last_angle=last_angle, has_obs=(rng.random(n) < 0.9).astype(np.uint8),
             Tcw=Tcw.reshape(-1), Tlw=Tlw.reshape(-1), fx=fx, fy=fy, cx=cx, cy=cy,
             mb=np.float32(bf / fx), mbf=bf, th=np.float32(th), mono=int(mono),
             check_orientation=int(check_orientation))
    return cur, p


# ----------------------------------------------------------------------------- local BA
def _rot_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        q = np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = np.zeros(4)
        q[i] = 0.5 * s
        s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s
        q[j] = (R[j, i] + R[i, j]) * s
        q[k] = (R[k, i] + R[i, k]) * s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def synth_lba_problem(seed: int, n_local: int = 20, n_fixed: int = 30, n_points: int = 4000,
                      obs_per_point: int = 6, cfg: str = "kitti", stereo_frac: float = 1.0,
                      outlier_frac: float = 0.05, include_kf0: bool = False):
    """Keyframes on an arc looking forward, points in a box ahead; float32 inputs like the reference
    (KeyFrame::GetPose / MapPoint::GetWorldPos are float32 cv::Mat).  SURVEY §8(d)."""
    rng = np.random.default_rng(4000 + seed)
    c = CONFIGS[cfg]
    fx, fy, cx, cy, bf = c["fx"], c["fy"], c["cx"], c["cy"], c["bf"]
    W, H = c["w"], c["h"]
    n_poses = n_local + n_fixed
    Twc_true = []
    for i in range(n_poses):
        s = i / max(n_poses - 1, 1)
        ang = 0.6 * s  # arc
        Rwc = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        twc = np.array([40.0 * np.sin(ang) * 0.8, 0.05 * np.sin(7 * s), 40.0 * s * 0.8])
        Twc_true.append((Rwc, twc))
    # points: sample in front of random keyframes
    pts = np.zeros((n_points, 3))
    for j in range(n_points):
        k = rng.integers(0, n_poses)
        Rwc, twc = Twc_true[k]
        z = rng.uniform(4.0, 40.0)
        u = rng.uniform(0, W)
        v = rng.uniform(0, H)
        Xc = np.array([(u - cx) / fx * z, (v - cy) / fy * z, z])
        pts[j] = Rwc @ Xc + twc
    sf = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2), np.float32)])).astype(np.float32)
    inv_sigma2_lv = (np.float32(1.0) / (sf * sf)).astype(np.float32)
    # order poses: local first then fixed (local = the last n_local keyframes, like a sliding window)
    order = list(range(n_fixed, n_poses)) + list(range(0, n_fixed))
    pose_id = np.array(order, dtype=np.int64) + (0 if include_kf0 else 1)
    pose_fixed = np.array([0] * n_local + [1] * n_fixed, dtype=np.uint8)
    if include_kf0:
        pose_fixed[pose_id == 0] = 1
    edge_pose, edge_point, edge_obs, edge_stereo, edge_is2 = [], [], [], [], []
    used_pts = []
    for j in range(n_points):
        vis = []
        for slot, k in enumerate(order):
            Rwc, twc = Twc_true[k]
            Xc = Rwc.T @ (pts[j] - twc)
            if Xc[2] < 1.0:
                continue
            u = fx * Xc[0] / Xc[2] + cx
            v = fy * Xc[1] / Xc[2] + cy
            if 0 <= u < W and 0 <= v < H:
                vis.append((slot, u, v, Xc[2]))
        if len(vis) < 2:
            continue
        sel = rng.permutation(len(vis))[:obs_per_point]
        sel.sort()
        if not any(vis[s][0] < n_local for s in sel):
            continue  # a local map point must be seen by a local keyframe
        pj = len(used_pts)
        used_pts.append(j)
        for s in sel:
            slot, u, v, z = vis[s]
            lvl = int(min(7, rng.geometric(0.4) - 1))
            sig = float(sf[lvl])
            out = rng.random() < outlier_frac
            du, dv = rng.normal(0, sig, 2)
            if out:
                du += rng.uniform(20, 80) * rng.choice([-1, 1])
                dv += rng.uniform(20, 80) * rng.choice([-1, 1])
            is_st = rng.random() < stereo_frac
            ur = (u + du) - bf / z + rng.normal(0, sig) if is_st else -1.0
            edge_pose.append(slot)
            edge_point.append(pj)
            edge_obs.append((np.float32(u + du), np.float32(v + dv), np.float32(ur)))
            edge_stereo.append(1 if is_st else 0)
            edge_is2.append(inv_sigma2_lv[lvl])
    used = np.array(used_pts, dtype=np.int64)
    P = pts[used]
    # perturb and convert to float32 inputs
    Tcw = np.zeros((n_poses, 4, 4), dtype=np.float32)
    for slot, k in enumerate(order):
        Rwc, twc = Twc_true[k]
        Rcw, tcw = Rwc.T, -Rwc.T @ twc
        if not pose_fixed[slot]:
            rv = rng.normal(0, 0.02 / np.sqrt(3), 3)
            th_ = np.linalg.norm(rv)
            K = np.array([[0, -rv[2], rv[1]], [rv[2], 0, -rv[0]], [-rv[1], rv[0], 0]])
            dR = np.eye(3) + np.sin(th_) / th_ * K + (1 - np.cos(th_)) / th_ ** 2 * K @ K
            Rcw = dR @ Rcw
            tcw = tcw + rng.normal(0, 0.1 / np.sqrt(3), 3)
        Tcw[slot, :3, :3] = Rcw
        Tcw[slot, :3, 3] = tcw
        Tcw[slot, 3, 3] = 1
    Pn = (P + rng.normal(0, 0.05, P.shape)).astype(np.float32)
    return dict(n_poses=n_poses, n_points=len(used), n_edges=len(edge_pose),
                pose_Tcw=Tcw.reshape(n_poses, 16), pose_fixed=pose_fixed, pose_id=pose_id,
                point_xyz=Pn, point_id=np.arange(len(used), dtype=np.int64) * 3 + 7,
                edge_pose=np.array(edge_pose, np.int32), edge_point=np.array(edge_point, np.int32),
                edge_obs=np.array(edge_obs, np.float32).reshape(-1, 3),
                edge_stereo=np.array(edge_stereo, np.uint8),
                edge_inv_sigma2=np.array(edge_is2, np.float32),
                fx=fx, fy=fy, cx=cx, cy=cy, bf=bf)


def lba_window_mix(seed: int, n_windows: int, hard_every: int = 0, hard=(0.5, 3.0, 2.0)):
    """Parameters of `n_windows` DIFFERENT LocalBundleAdjustment windows like LocalMapping meets them (src/Optimizer.cc:457-505:
    the local keyframes are the covisibility neighbourhood of the new keyframe, the fixed ones whatever else sees its points):
    10-40 local keyframes, 0.5-1.5 times as many fixed ones, 3 000-9 000 candidate points (about 70 % of them end up in the
    window: 2-6 k), 4-8 observations per point, 5-20 % gross outliers.  hard_every = k: every k-th window starts far from the optimum
    (perturb_lba_problem: steps get rejected, some optimisations stop early).  Keyword dicts for synth_lba_problem() / _lba_from_kwargs()."""
    rng = np.random.default_rng(31000 + seed)
    mix = []
    for i in range(n_windows):
        n_local = int(rng.integers(10, 41))
        n_fixed = int(np.clip(round(n_local * rng.uniform(0.5, 1.5)), 2, 45))
        mix.append(dict(seed=100000 + 1000 * seed + i, n_local=n_local, n_fixed=n_fixed, n_points=int(rng.integers(3000, 9001)),
                        obs_per_point=int(rng.integers(4, 9)), outlier_frac=float(rng.uniform(0.05, 0.20))))
        if hard_every and i % hard_every == hard_every - 1:   # a window that starts off the optimum: rejected steps (perturb_lba_problem)
            mix[-1]["hard"] = tuple(hard)
            mix[-1]["stereo_frac"] = 0.3   # (mostly monocular observations: depth is what a bad start gets wrong)
    return mix


def perturb_lba_problem(prob, seed, rot_sigma, trans_sigma, point_sigma):
    """the free keyframes and the points of a window moved away from the optimum (a rotation of rot_sigma rad about a random axis,
    Gaussian offsets): Levenberg-Marquardt then rejects steps (lambda grows, the estimates are restored) and may give up"""
    def rot(axis, a):
        axis = axis / np.linalg.norm(axis)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K
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


def _lba_from_kwargs(kw):
    kw = dict(kw)
    hard = kw.pop("hard", None)
    prob = synth_lba_problem(**kw)
    return perturb_lba_problem(prob, kw["seed"], *hard) if hard else prob


def synth_lba_problems(mix):
    """synth_lba_problem(**kw) for every kw of `mix` (worker processes when there are many: scenario.pmap)"""
    from . import scenario
    return scenario.pmap(_lba_from_kwargs, list(mix))


def tcw_to_qt(Tcw16):
    """Converter::toSE3Quat (src/Converter.cc:37-47): float32 4x4 -> (qx qy qz qw tx ty tz) double."""
    T = np.asarray(Tcw16, dtype=np.float32).reshape(4, 4).astype(np.float64)
    q = _rot_to_quat(T[:3, :3])
    return np.concatenate([q, T[:3, 3]])


def synth_pose_problem(seed: int, n: int = 800, cfg: str = "kitti", stereo_frac: float = 0.8,
                       outlier_frac: float = 0.1, rot_err: float = 0.01, trans_err: float = 0.05):
    """Frame with n map-point matches for Optimizer::PoseOptimization (src/Optimizer.cc:239-452)."""
    rng = np.random.default_rng(5000 + seed)
    c = CONFIGS[cfg]
    fx, fy, cx, cy, bf = c["fx"], c["fy"], c["cx"], c["cy"], c["bf"]
    W, H = c["w"], c["h"]
    ang = rng.uniform(-0.3, 0.3)
    Rcw = np.array([[np.cos(ang), 0, -np.sin(ang)], [0, 1, 0], [np.sin(ang), 0, np.cos(ang)]])
    tcw = rng.uniform(-1, 1, 3)
    z = rng.uniform(3.0, 40.0, n)
    u = rng.uniform(20, W - 20, n)
    v = rng.uniform(20, H - 20, n)
    Xc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], axis=1)
    Xw = (Xc - tcw) @ Rcw  # Rcw^T (Xc - tcw)
    sf = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2), np.float32)])).astype(np.float32)
    lvl = np.minimum(rng.geometric(0.4, n) - 1, 7)
    sig = sf[lvl].astype(np.float64)
    du, dv, dr = rng.normal(0, 1, n) * sig, rng.normal(0, 1, n) * sig, rng.normal(0, 1, n) * sig
    out = rng.random(n) < outlier_frac
    du[out] += rng.uniform(15, 60, int(out.sum())) * rng.choice([-1, 1], int(out.sum()))
    dv[out] += rng.uniform(15, 60, int(out.sum())) * rng.choice([-1, 1], int(out.sum()))
    stereo = (rng.random(n) < stereo_frac)
    ur = np.where(stereo, u + du - bf / z + dr, -1.0)
    obs = np.stack([u + du, v + dv, ur], axis=1).astype(np.float32)
    rv = rng.normal(0, rot_err / np.sqrt(3), 3)
    th_ = np.linalg.norm(rv)
    K = np.array([[0, -rv[2], rv[1]], [rv[2], 0, -rv[0]], [-rv[1], rv[0], 0]])
    dR = np.eye(3) + np.sin(th_) / th_ * K + (1 - np.cos(th_)) / th_ ** 2 * K @ K
    T = np.eye(4)
    T[:3, :3] = dR @ Rcw
    T[:3, 3] = tcw + rng.normal(0, trans_err / np.sqrt(3), 3)
    return dict(n=n, Xw=Xw.astype(np.float32), obs=obs, stereo=stereo.astype(np.uint8),
                inv_sigma2=(np.float32(1.0) / (sf[lvl] * sf[lvl])).astype(np.float32),
                fx=fx, fy=fy, cx=cx, cy=cy, bf=bf, Tcw=T.astype(np.float32).reshape(16))

"""ctypes binding of the ORACLE library (test infrastructure only; PARITY UNPINNED).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborb_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_extractor_create.restype = C.c_void_p
        L.orc_extractor_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_extractor_destroy.argtypes = [C.c_void_p]
        for name in ("scale_factors", "inv_scale_factors", "sigma2", "inv_sigma2"):
            f = getattr(L, "orc_extractor_" + name)
            f.restype = C.POINTER(C.c_float)
            f.argtypes = [C.c_void_p]
        for name in ("features_per_level", "umax"):
            f = getattr(L, "orc_extractor_" + name)
            f.restype = C.POINTER(C.c_int)
            f.argtypes = [C.c_void_p]
        L.orc_extractor_levels.argtypes = [C.c_void_p]
        L.orc_extractor_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_extractor_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_extractor_level_plane.restype = C.c_void_p
        L.orc_extractor_level_plane.argtypes = [C.c_void_p, C.c_int]
        L.orc_extractor_level_blurred.restype = C.c_void_p
        L.orc_extractor_level_blurred.argtypes = [C.c_void_p, C.c_int]
        L.orc_extractor_level_candidates.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p),
                                                     C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.orc_extractor_level_nkeys.argtypes = [C.c_void_p, C.c_int]
        L.orc_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_int]
        L.orc_fast_corner_score.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_cv_round_f.argtypes = [C.c_float]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_gaussian_blur7_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_copy_make_border_reflect101.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_distribute_octree.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_search_by_bow.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_search_by_projection_mp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_search_by_projection_last.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_lba_solve.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_search_by_bow_kf.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_fuse.argtypes = [C.c_void_p] * 4
        L.orc_assign_features_to_grid.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_stereo_from_rgbd.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_stereo_from_rgbd.restype = None
        L.orc_is_in_frustum.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float] + [C.c_void_p] * 6
        L.orc_is_in_frustum.restype = None
        L.orc_search_for_initialization.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                    C.c_float, C.c_int, C.c_void_p]
        L.orc_fuse_sim3.argtypes = [C.c_void_p] * 4
        L.orc_search_by_projection_kf.argtypes = [C.c_void_p] * 3
        L.orc_search_by_sim3.argtypes = [C.c_void_p] * 5
        L.orc_search_by_projection_reloc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_search_for_triangulation.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_compute_distinctive_descriptors.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_compute_distinctive_descriptors.restype = None
        L.orc_vocab_create.restype = C.c_void_p
        L.orc_vocab_destroy.argtypes = [C.c_void_p]
        for name in ("k", "L", "scoring", "weighting", "nodes", "size"):
            getattr(L, "orc_vocab_" + name).argtypes = [C.c_void_p]
        L.orc_vocab_set_nodes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4
        L.orc_vocab_load_binary.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_vocab_save_binary.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_vocab_load_text.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_vocab_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 9
        L.orc_vocab_score_l1.restype = C.c_double
        L.orc_vocab_score_l1.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_compute_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_se3_exp.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_se3_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_quat_from_rot.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_rot_from_quat.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pose_from_Tcw_f32.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pose_to_Tcw_f32.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_edge_linearize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_double] * 5 + [C.c_void_p] * 3
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Extractor:
    """ORBextractor restatement (reference src/ORBextractor.cc)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.h = self.L.orc_extractor_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        if not self.h:
            raise ValueError("bad extractor parameters")
        self.nlevels = nlevels
        self.nfeatures = nfeatures

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_extractor_destroy(self.h)
            self.h = None

    def _farr(self, name, n, ct=np.float32):
        ptr = getattr(self.L, "orc_extractor_" + name)(self.h)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(ct).copy()

    @property
    def scale_factors(self):
        return self._farr("scale_factors", self.nlevels)

    @property
    def inv_scale_factors(self):
        return self._farr("inv_scale_factors", self.nlevels)

    @property
    def sigma2(self):
        return self._farr("sigma2", self.nlevels)

    @property
    def inv_sigma2(self):
        return self._farr("inv_sigma2", self.nlevels)

    @property
    def features_per_level(self):
        return self._farr("features_per_level", self.nlevels, np.int32)

    @property
    def umax(self):
        return self._farr("umax", 16, np.int32)

    def extract(self, img: np.ndarray, cap: int | None = None):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        cap = cap or (self.nfeatures * 2 + 64)
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = C.c_int(0)
        st = self.L.orc_extractor_extract(self.h, _p(img), w, h, img.strides[0], _p(kps), _p(desc), cap, C.byref(n))
        if st == -2 and n.value > cap:   # DistributeOctTree returned more than fits (wide images: up to 4 * round(W / H) extra per level): once more with room
            return self.extract(img, cap=n.value)
        if st != 0:
            raise RuntimeError(f"oracle extract failed: {st}")
        return kps[: n.value].copy(), desc[: n.value].copy()

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        pitch = self.L.orc_extractor_level_size(self.h, level, C.byref(w), C.byref(h))
        return w.value, h.value, pitch

    def level_plane(self, level):
        w, h, pitch = self.level_size(level)
        ptr = self.L.orc_extractor_level_plane(self.h, level)
        buf = (C.c_uint8 * (pitch * h)).from_address(ptr)
        a = np.frombuffer(buf, dtype=np.uint8, count=pitch * (h - 1) + w)
        return np.lib.stride_tricks.as_strided(a, shape=(h, w), strides=(pitch, 1)).copy()

    def level_blurred(self, level):
        w, h, _ = self.level_size(level)
        ptr = self.L.orc_extractor_level_blurred(self.h, level)
        buf = (C.c_uint8 * (w * h)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint8).reshape(h, w).copy()

    def level_candidates(self, level):
        xs, ys, sc = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n = self.L.orc_extractor_level_candidates(self.h, level, C.byref(xs), C.byref(ys), C.byref(sc))
        if n <= 0:
            return (np.zeros(0, np.int16),) * 2 + (np.zeros(0, np.uint8),)
        x = np.frombuffer((C.c_int16 * n).from_address(xs.value), dtype=np.int16).copy()
        y = np.frombuffer((C.c_int16 * n).from_address(ys.value), dtype=np.int16).copy()
        s = np.frombuffer((C.c_uint8 * n).from_address(sc.value), dtype=np.uint8).copy()
        return x, y, s

    def level_nkeys(self, level):
        return self.L.orc_extractor_level_nkeys(self.h, level)


def fast9_16(img, threshold):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    cap = w * h
    xs = np.zeros(cap, np.int16)
    ys = np.zeros(cap, np.int16)
    sc = np.zeros(cap, np.uint8)
    n = lib().orc_fast9_16(_p(img), w, h, img.strides[0], threshold, _p(xs), _p(ys), _p(sc), cap)
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def fast_corner_score(img, x, y, threshold):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    addr = img.ctypes.data + y * img.strides[0] + x
    return lib().orc_fast_corner_score(C.c_void_p(addr), img.strides[0], threshold)


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(float(y), float(x)))


def cv_round(v):
    return int(lib().orc_cv_round_f(float(v)))


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw = src.shape
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), sw, sh, src.strides[0], _p(dst), dw, dh, dw)
    return dst


def gaussian_blur7(src):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape
    dst = np.zeros_like(src)
    lib().orc_gaussian_blur7_u8(_p(src), w, h, src.strides[0], _p(dst), w)
    return dst


def copy_make_border(src, border=19):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape
    dst = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
    lib().orc_copy_make_border_reflect101(_p(src), w, h, src.strides[0], _p(dst), border, w + 2 * border)
    return dst


def distribute_octree(xs, ys, resp, minX, maxX, minY, maxY, N):
    xs = np.ascontiguousarray(xs, np.float32)
    ys = np.ascontiguousarray(ys, np.float32)
    resp = np.ascontiguousarray(resp, np.float32)
    n = len(xs)
    out = np.zeros(max(n, 1), np.int32)
    k = lib().orc_distribute_octree(_p(xs), _p(ys), _p(resp), n, minX, maxX, minY, maxY, N, _p(out), len(out))
    return out[:k].copy()


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


# ---------------------------------------------------------------------------- matcher structs
class _BowProblem(C.Structure):
    _fields_ = [("n_kf", C.c_int), ("n_f", C.c_int), ("desc_kf", C.c_void_p), ("desc_f", C.c_void_p),
                ("kf_has_mp", C.c_void_p), ("angle_kf", C.c_void_p), ("angle_f", C.c_void_p),
                ("n_nodes_kf", C.c_int), ("n_nodes_f", C.c_int),
                ("node_id_kf", C.c_void_p), ("node_off_kf", C.c_void_p), ("node_idx_kf", C.c_void_p),
                ("node_id_f", C.c_void_p), ("node_off_f", C.c_void_p), ("node_idx_f", C.c_void_p),
                ("nnratio", C.c_float), ("check_orientation", C.c_int)]


class _FrameView(C.Structure):
    _fields_ = [("n_f", C.c_int), ("desc_f", C.c_void_p), ("kp_x", C.c_void_p), ("kp_y", C.c_void_p),
                ("kp_octave", C.c_void_p), ("kp_angle", C.c_void_p), ("u_right", C.c_void_p),
                ("scale_factors", C.c_void_p), ("n_levels", C.c_int),
                ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("grid_w_inv", C.c_float), ("grid_h_inv", C.c_float),
                ("grid_off", C.c_void_p), ("grid_idx", C.c_void_p), ("f_mp_state", C.c_void_p)]


class _ProjMp(C.Structure):
    _fields_ = [("n_mp", C.c_int), ("track_in_view", C.c_void_p), ("pred_level", C.c_void_p),
                ("view_cos", C.c_void_p), ("proj_x", C.c_void_p), ("proj_y", C.c_void_p),
                ("proj_xr", C.c_void_p), ("desc", C.c_void_p), ("has_obs", C.c_void_p),
                ("th", C.c_float), ("nnratio", C.c_float)]


class _ProjLast(C.Structure):
    _fields_ = [("n_last", C.c_int), ("last_valid", C.c_void_p), ("world_pos", C.c_void_p),
                ("desc", C.c_void_p), ("last_octave", C.c_void_p), ("last_angle", C.c_void_p),
                ("has_obs", C.c_void_p), ("Tcw", C.c_float * 16), ("Tlw", C.c_float * 16),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("mb", C.c_float), ("mbf", C.c_float), ("th", C.c_float),
                ("mono", C.c_int), ("check_orientation", C.c_int)]


def _fill(struct, d, keep):
    for name, ct in struct._fields_:
        if name not in d:
            continue
        v = d[name]
        if isinstance(v, np.ndarray) and v.ndim == 0:
            v = v.item()
        if isinstance(v, np.ndarray) and ct is C.c_void_p:
            v = np.ascontiguousarray(v)
            keep.append(v)
            setattr(struct, name, v.ctypes.data)
        elif isinstance(v, np.ndarray):
            arr = ct(*[float(x) for x in v.reshape(-1)]) if hasattr(ct, "_length_") else (C.c_float * 16)(*[float(x) for x in v.reshape(-1)])
            setattr(struct, name, arr)
        else:
            setattr(struct, name, v.item() if hasattr(v, "item") else v)


def search_by_bow(p: dict):
    keep = []
    s = _BowProblem()
    d = dict(p)
    d["n_kf"] = len(p["desc_kf"])
    d["n_f"] = len(p["desc_f"])
    d["n_nodes_kf"] = len(p["node_id_kf"])
    d["n_nodes_f"] = len(p["node_id_f"])
    _fill(s, d, keep)
    match = np.zeros(d["n_f"], np.int32)
    n = lib().orc_search_by_bow(C.byref(s), _p(match))
    return n, match


def _frame_view(f: dict, keep):
    s = _FrameView()
    _fill(s, f, keep)
    return s


def search_by_projection_mp(f: dict, mp: dict):
    keep = []
    fv = _frame_view(f, keep)
    s = _ProjMp()
    _fill(s, mp, keep)
    match = np.zeros(f["n_f"], np.int32)
    n = lib().orc_search_by_projection_mp(C.byref(fv), C.byref(s), _p(match))
    return n, match


def search_by_projection_last(cur: dict, p: dict):
    keep = []
    fv = _frame_view(cur, keep)
    s = _ProjLast()
    _fill(s, p, keep)
    match = np.zeros(cur["n_f"], np.int32)
    n = lib().orc_search_by_projection_last(C.byref(fv), C.byref(s), _p(match))
    return n, match


# ---------------------------------------------------------------------------- local BA
class _ProjGen(C.Structure):
    _fields_ = [("n_pts", C.c_int), ("valid", C.c_void_p), ("pos", C.c_void_p), ("max_dist", C.c_void_p),
                ("min_dist", C.c_void_p), ("normal", C.c_void_p), ("desc", C.c_void_p), ("q_angle", C.c_void_p),
                ("R", C.c_float * 9), ("t", C.c_float * 3), ("Ow", C.c_float * 3), ("R2", C.c_float * 9),
                ("t2", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("bf", C.c_float), ("log_scale_factor", C.c_float), ("inv_level_sigma2", C.c_void_p), ("th", C.c_float)]


def _proj_gen(p, keep):
    s = _ProjGen()
    _fill(s, p, keep)
    return s


def fuse(f: dict, p: dict, sim3=False):
    """Fuse search part (src/ORBmatcher.cc:825-975; sim3=True: :977-1100) -> (nFused, best_idx, best_dist)"""
    keep = []
    fv, pg = _frame_view(f, keep), _proj_gen(p, keep)
    bi, bd = np.zeros(max(p["n_pts"], 1), np.int32), np.zeros(max(p["n_pts"], 1), np.int32)
    fn = lib().orc_fuse_sim3 if sim3 else lib().orc_fuse
    n = fn(C.byref(fv), C.byref(pg), _p(bi), _p(bd))
    return n, bi[: p["n_pts"]], bd[: p["n_pts"]]


def search_by_projection_kf(f: dict, p: dict):
    """SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) src/ORBmatcher.cc:290-403 -> (nmatches, match_f)"""
    keep = []
    fv, pg = _frame_view(f, keep), _proj_gen(p, keep)
    m = np.zeros(max(f["n_f"], 1), np.int32)
    n = lib().orc_search_by_projection_kf(C.byref(fv), C.byref(pg), _p(m))
    return n, m[: f["n_f"]]


def search_by_sim3(f1: dict, f2: dict, p12: dict, p21: dict):
    """SearchBySim3 src/ORBmatcher.cc:1102-1326 -> (nFound, match12)"""
    keep = []
    a, b, c, d = _frame_view(f1, keep), _frame_view(f2, keep), _proj_gen(p12, keep), _proj_gen(p21, keep)
    m = np.zeros(max(p12["n_pts"], 1), np.int32)
    n = lib().orc_search_by_sim3(C.byref(a), C.byref(b), C.byref(c), C.byref(d), _p(m))
    return n, m[: p12["n_pts"]]


def search_by_projection_reloc(f: dict, p: dict, orb_dist=100, check_orientation=True):
    """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) src/ORBmatcher.cc:1472-1599"""
    keep = []
    fv, pg = _frame_view(f, keep), _proj_gen(p, keep)
    m = np.zeros(max(f["n_f"], 1), np.int32)
    n = lib().orc_search_by_projection_reloc(C.byref(fv), C.byref(pg), int(orb_dist), int(check_orientation), _p(m))
    return n, m[: f["n_f"]]


def assign_features_to_grid(kp_x, kp_y, min_x, min_y, grid_w_inv, grid_h_inv):
    """Frame::AssignFeaturesToGrid src/Frame.cc:259-274 -> (grid_off[3073], grid_idx)"""
    kp_x, kp_y = np.ascontiguousarray(kp_x, np.float32), np.ascontiguousarray(kp_y, np.float32)
    off = np.zeros(64 * 48 + 1, np.int32)
    idx = np.zeros(max(len(kp_x), 1), np.int32)
    n = lib().orc_assign_features_to_grid(len(kp_x), _p(kp_x), _p(kp_y), np.float32(min_x), np.float32(min_y),
                                          np.float32(grid_w_inv), np.float32(grid_h_inv), _p(off), _p(idx))
    return off, idx[:n]


def stereo_from_rgbd(kp_x, kp_y, kpun_x, depth_img, mbf):
    """Frame::ComputeStereoFromRGBD src/Frame.cc:672-693 -> (mvuRight, mvDepth)"""
    kp_x, kp_y, kpun_x = (np.ascontiguousarray(a, np.float32) for a in (kp_x, kp_y, kpun_x))
    depth_img = np.ascontiguousarray(depth_img, np.float32)
    ur, dp = np.zeros(max(len(kp_x), 1), np.float32), np.zeros(max(len(kp_x), 1), np.float32)
    lib().orc_stereo_from_rgbd(len(kp_x), _p(kp_x), _p(kp_y), _p(kpun_x), _p(depth_img), depth_img.shape[1], np.float32(mbf),
                               _p(ur), _p(dp))
    return ur[: len(kp_x)], dp[: len(kp_x)]


def undistort_points(xy, fx, fy, cx, cy, dist):
    """cv::undistortPoints(xy, K, dist, noArray(), K) (Frame::UndistortKeyPoints / ComputeImageBounds, src/Frame.cc:433-493);
    dist = k1 k2 p1 p2 k3"""
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    d = np.zeros(5, np.float32)
    d[: len(dist)] = np.asarray(dist, np.float32)
    out = np.zeros_like(xy)
    L = lib()
    L.orc_undistort_points.argtypes = [C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_undistort_points(len(xy), _p(xy), np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy), _p(d), _p(out))
    return out


def is_in_frustum(f: dict, p: dict, viewing_cos_limit=0.5):
    """Frame::isInFrustum (src/Frame.cc:298-354) for all points of p against frame view f -> dict of the track members"""
    keep = []
    pg = _proj_gen(p, keep)
    n = p["n_pts"]
    iv = np.zeros(max(n, 1), np.uint8)
    px, py, pr, vc = (np.zeros(max(n, 1), np.float32) for _ in range(4))
    lv = np.zeros(max(n, 1), np.int32)
    lib().orc_is_in_frustum(C.byref(pg), np.float32(f["min_x"]), np.float32(f["max_x"]), np.float32(f["min_y"]),
                            np.float32(f["max_y"]), int(f["n_levels"]), np.float32(viewing_cos_limit), _p(iv), _p(px), _p(py),
                            _p(pr), _p(lv), _p(vc))
    return dict(track_in_view=iv[:n], proj_x=px[:n], proj_y=py[:n], proj_xr=pr[:n], pred_level=lv[:n], view_cos=vc[:n])


def search_for_initialization(f2: dict, q: dict, window_size=100, nnratio=0.9, check_orientation=True):
    """SearchForInitialization src/ORBmatcher.cc:405-520 -> (nmatches, vnMatches12)"""
    keep = []
    fv = _frame_view(f2, keep)
    d1 = np.ascontiguousarray(q["desc1"], np.uint8)
    o1 = np.ascontiguousarray(q["octave1"], np.int32)
    a1 = np.ascontiguousarray(q["angle1"], np.float32)
    pv = np.ascontiguousarray(q["prev_xy"], np.float32)
    m = np.zeros(max(len(d1), 1), np.int32)
    n = lib().orc_search_for_initialization(C.byref(fv), len(d1), _p(d1), _p(o1), _p(a1), _p(pv), int(window_size),
                                            np.float32(nnratio), int(check_orientation), _p(m))
    return n, m[: len(d1)]


class _BowKfProblem(C.Structure):
    _fields_ = [("n1", C.c_int), ("n2", C.c_int), ("desc1", C.c_void_p), ("desc2", C.c_void_p),
                ("has_mp1", C.c_void_p), ("has_mp2", C.c_void_p), ("angle1", C.c_void_p), ("angle2", C.c_void_p),
                ("n_nodes1", C.c_int), ("n_nodes2", C.c_int),
                ("node_id1", C.c_void_p), ("node_off1", C.c_void_p), ("node_idx1", C.c_void_p),
                ("node_id2", C.c_void_p), ("node_off2", C.c_void_p), ("node_idx2", C.c_void_p),
                ("nnratio", C.c_float), ("check_orientation", C.c_int)]


class _TriangProblem(C.Structure):
    _fields_ = [("n1", C.c_int), ("n2", C.c_int), ("desc1", C.c_void_p), ("desc2", C.c_void_p),
                ("has_mp1", C.c_void_p), ("has_mp2", C.c_void_p),
                ("x1", C.c_void_p), ("y1", C.c_void_p), ("angle1", C.c_void_p), ("u_right1", C.c_void_p),
                ("x2", C.c_void_p), ("y2", C.c_void_p), ("angle2", C.c_void_p), ("u_right2", C.c_void_p),
                ("octave2", C.c_void_p), ("scale_factors2", C.c_void_p), ("level_sigma2_2", C.c_void_p),
                ("F12", C.c_float * 9), ("ex", C.c_float), ("ey", C.c_float),
                ("only_stereo", C.c_int), ("check_orientation", C.c_int),
                ("n_nodes1", C.c_int), ("n_nodes2", C.c_int),
                ("node_id1", C.c_void_p), ("node_off1", C.c_void_p), ("node_idx1", C.c_void_p),
                ("node_id2", C.c_void_p), ("node_off2", C.c_void_p), ("node_idx2", C.c_void_p)]


def _kfkf(struct_t, fn, p):
    keep = []
    s = struct_t()
    d = dict(p)
    d["n1"], d["n2"] = len(p["desc1"]), len(p["desc2"])
    d["n_nodes1"], d["n_nodes2"] = len(p["node_id1"]), len(p["node_id2"])
    _fill(s, d, keep)
    match = np.zeros(max(d["n1"], 1), np.int32)
    n = fn(C.byref(s), _p(match))
    return n, match[: d["n1"]]


def search_by_bow_kf(p: dict):
    """SearchByBoW(pKF1, pKF2, vpMatches12) src/ORBmatcher.cc:522-655 -> (nmatches, match12)"""
    return _kfkf(_BowKfProblem, lib().orc_search_by_bow_kf, p)


def search_for_triangulation(p: dict):
    """SearchForTriangulation src/ORBmatcher.cc:657-823 -> (nmatches, vMatches12)"""
    return _kfkf(_TriangProblem, lib().orc_search_for_triangulation, p)


def compute_distinctive_descriptors(off, desc):
    """MapPoint::ComputeDistinctiveDescriptors src/MapPoint.cc:275-340 for a batch (CSR) -> best index per point"""
    off = np.ascontiguousarray(off, np.int32)
    desc = np.ascontiguousarray(desc, np.uint8)
    best = np.zeros(max(len(off) - 1, 1), np.int32)
    lib().orc_compute_distinctive_descriptors(len(off) - 1, _p(off), _p(desc), _p(best))
    return best[: len(off) - 1]


class _StereoProblem(C.Structure):
    _fields_ = [("n_left", C.c_int), ("n_right", C.c_int), ("kp_left", C.c_void_p), ("kp_right", C.c_void_p),
                ("desc_left", C.c_void_p), ("desc_right", C.c_void_p), ("n_levels", C.c_int),
                ("scale_factors", C.c_void_p), ("inv_scale_factors", C.c_void_p),
                ("left_planes", C.c_void_p), ("right_planes", C.c_void_p),
                ("left_pitch", C.c_void_p), ("right_pitch", C.c_void_p), ("level_w", C.c_void_p),
                ("level_h", C.c_void_p), ("mb", C.c_float), ("mbf", C.c_float)]


def compute_stereo_matches(ext_left: "Extractor", ext_right: "Extractor", kps_l, desc_l, kps_r, desc_r,
                           mb: float, mbf: float):
    """Frame::ComputeStereoMatches (src/Frame.cc:495-669) on the pyramids the two oracle extractors hold
    from their last extract().  Returns (mvuRight, mvDepth, n_before_cull)."""
    L = lib()
    nl = ext_left.nlevels
    kps_l = np.ascontiguousarray(kps_l, dtype=KP_DTYPE)
    kps_r = np.ascontiguousarray(kps_r, dtype=KP_DTYPE)
    desc_l = np.ascontiguousarray(desc_l, dtype=np.uint8)
    desc_r = np.ascontiguousarray(desc_r, dtype=np.uint8)
    sf = ext_left.scale_factors
    isf = ext_left.inv_scale_factors
    pl = (C.c_void_p * nl)(*[L.orc_extractor_level_plane(ext_left.h, l) for l in range(nl)])
    pr = (C.c_void_p * nl)(*[L.orc_extractor_level_plane(ext_right.h, l) for l in range(nl)])
    szl = [ext_left.level_size(l) for l in range(nl)]
    szr = [ext_right.level_size(l) for l in range(nl)]
    lw = np.array([s[0] for s in szl], np.int32)
    lh = np.array([s[1] for s in szl], np.int32)
    lp = np.array([s[2] for s in szl], np.int32)
    rp = np.array([s[2] for s in szr], np.int32)
    P = _StereoProblem(len(kps_l), len(kps_r), _p(kps_l).value, _p(kps_r).value, _p(desc_l).value, _p(desc_r).value,
                       nl, _p(sf).value, _p(isf).value, C.cast(pl, C.c_void_p).value, C.cast(pr, C.c_void_p).value,
                       _p(lp).value, _p(rp).value, _p(lw).value, _p(lh).value, np.float32(mb), np.float32(mbf))
    ur = np.zeros(len(kps_l), np.float32)
    dp = np.zeros(len(kps_l), np.float32)
    n = L.orc_compute_stereo_matches(C.byref(P), _p(ur), _p(dp))
    return ur, dp, n


class Vocabulary:
    """ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> restatement (dbow_oracle.c)."""

    def __init__(self):
        self.L_ = lib()
        self.h = self.L_.orc_vocab_create()

    def __del__(self):
        if getattr(self, "h", None):
            self.L_.orc_vocab_destroy(self.h)
            self.h = None

    def set_nodes(self, k, L, scoring, weighting, parent, desc, weight, is_leaf):
        parent = np.ascontiguousarray(parent, np.int32)
        desc = np.ascontiguousarray(desc, np.uint8)
        weight = np.ascontiguousarray(weight, np.float64)
        is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        st = self.L_.orc_vocab_set_nodes(self.h, k, L, scoring, weighting, len(parent), _p(parent), _p(desc), _p(weight),
                                         _p(is_leaf))
        if st:
            raise ValueError(f"set_nodes failed: {st}")

    def load_binary(self, path):
        return self.L_.orc_vocab_load_binary(self.h, str(path).encode()) == 0

    def save_binary(self, path):
        return self.L_.orc_vocab_save_binary(self.h, str(path).encode()) == 0

    def load_text(self, path):
        return self.L_.orc_vocab_load_text(self.h, str(path).encode()) == 0

    def info(self):
        g = lambda n: getattr(self.L_, "orc_vocab_" + n)(self.h)  # noqa: E731
        return dict(k=g("k"), L=g("L"), scoring=g("scoring"), weighting=g("weighting"), nodes=g("nodes"), words=g("size"))

    def transform(self, desc, levelsup=4):
        """-> dict(bow_word, bow_value, fv_node, fv_off, fv_idx, word_of, node_of) or None if the vocabulary is empty"""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        bw = np.zeros(max(n, 1), np.uint32)
        bv = np.zeros(max(n, 1), np.float64)
        fn = np.zeros(max(n, 1), np.int32)
        fo = np.zeros(n + 2, np.int32)
        fi = np.zeros(max(n, 1), np.int32)
        wo = np.zeros(max(n, 1), np.uint32)
        no = np.zeros(max(n, 1), np.uint32)
        nb, nf = C.c_int(0), C.c_int(0)
        st = self.L_.orc_vocab_transform(self.h, _p(desc), n, levelsup, _p(bw), _p(bv), C.byref(nb), _p(fn), _p(fo), _p(fi),
                                         C.byref(nf), _p(wo), _p(no))
        if st:
            return None
        nb, nf = nb.value, nf.value
        return dict(bow_word=bw[:nb].copy(), bow_value=bv[:nb].copy(), fv_node=fn[:nf].copy(), fv_off=fo[: nf + 1].copy(),
                    fv_idx=fi[: fo[nf]].copy(), word_of=wo[:n].copy(), node_of=no[:n].copy())


def vocab_score_l1(a, b):
    L = lib()
    w1, v1 = np.ascontiguousarray(a["bow_word"], np.uint32), np.ascontiguousarray(a["bow_value"], np.float64)
    w2, v2 = np.ascontiguousarray(b["bow_word"], np.uint32), np.ascontiguousarray(b["bow_value"], np.float64)
    return L.orc_vocab_score_l1(_p(w1), _p(v1), len(w1), _p(w2), _p(v2), len(w2))


class _LbaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int), ("n_points", C.c_int), ("n_edges", C.c_int),
                ("pose_qt", C.c_void_p), ("pose_fixed", C.c_void_p), ("pose_id", C.c_void_p),
                ("point_xyz", C.c_void_p), ("point_id", C.c_void_p),
                ("edge_pose", C.c_void_p), ("edge_point", C.c_void_p), ("edge_obs", C.c_void_p),
                ("edge_stereo", C.c_void_p), ("edge_inv_sigma2", C.c_void_p),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("bf", C.c_double), ("stop_flag", C.c_void_p), ("iters1", C.c_int), ("iters2", C.c_int),
                ("stop_at_poll", C.c_int)]


class _LbaResult(C.Structure):
    _fields_ = [("edge_chi2", C.c_void_p), ("edge_depth_pos", C.c_void_p), ("edge_outlier", C.c_void_p),
                ("edge_level1", C.c_void_p), ("lambda_trace", C.c_double * 64),
                ("chi2_trace", C.c_double * 64), ("n_trace", C.c_int),
                ("iters_done1", C.c_int), ("iters_done2", C.c_int), ("polls", C.c_int), ("stop_poll", C.c_int),
                ("trials", C.c_int)]


def pose_from_Tcw(T16):
    T = np.ascontiguousarray(T16, np.float32).reshape(16)
    qt = np.zeros(7, np.float64)
    lib().orc_pose_from_Tcw_f32(_p(T), _p(qt))
    return qt


def pose_to_Tcw(qt):
    qt = np.ascontiguousarray(qt, np.float64)
    T = np.zeros(16, np.float32)
    lib().orc_pose_to_Tcw_f32(_p(qt), _p(T))
    return T


_CONTRACTED = None


def contracted_lib():
    """liborb_oracle_contracted.so: the same sources with the reference's own g2o flags (oracle/Makefile "contracted": fused
    multiply-adds).  Measurement of the restatement's numerical resolution only (parity.lba_resolution); a test's expected value
    always comes from lib()."""
    global _CONTRACTED
    if _CONTRACTED is None:
        path = os.path.join(_HERE, "liborb_oracle_contracted.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE, "contracted"])
        _CONTRACTED = C.CDLL(path)
    return _CONTRACTED


def lba_solve(prob: dict, iters1=5, iters2=10, stop_flag=None, stop_at_poll=0, variant=0, contracted=False):
    """Runs the LocalBA numerical core on a synth_lba_problem()-style dict with float32 inputs.
    Returns dict with float32 write-back poses/points like Optimizer.cc:763-778.
    variant / contracted (measurement only, lba_oracle.c orc_set_lba_variant / contracted_lib()): the same algorithm with its
    arithmetic re-associated at rounding level."""
    n_poses, n_points, n_edges = prob["n_poses"], prob["n_points"], prob["n_edges"]
    qt = np.stack([pose_from_Tcw(prob["pose_Tcw"][i]) for i in range(n_poses)]).astype(np.float64)
    pts = np.ascontiguousarray(prob["point_xyz"], np.float32).astype(np.float64)  # Converter::toVector3d
    obs = np.ascontiguousarray(prob["edge_obs"], np.float32).astype(np.float64)
    keep = [qt, pts, obs]
    s = _LbaProblem()
    s.n_poses, s.n_points, s.n_edges = n_poses, n_points, n_edges
    s.pose_qt = qt.ctypes.data
    s.point_xyz = pts.ctypes.data
    s.edge_obs = obs.ctypes.data
    for name, dt in (("pose_fixed", np.uint8), ("pose_id", np.int64), ("point_id", np.int64),
                     ("edge_pose", np.int32), ("edge_point", np.int32), ("edge_stereo", np.uint8),
                     ("edge_inv_sigma2", np.float32)):
        a = np.ascontiguousarray(prob[name], dt)
        keep.append(a)
        setattr(s, name, a.ctypes.data)
    # KeyFrame::fx..mbf are float members copied into the edges' double fields (Optimizer.cc:613-616,642-646)
    s.fx, s.fy, s.cx, s.cy, s.bf = (float(np.float32(prob[k])) for k in ("fx", "fy", "cx", "cy", "bf"))
    if stop_flag is not None:
        keep.append(stop_flag)
        s.stop_flag = stop_flag.ctypes.data
    s.iters1, s.iters2 = iters1, iters2
    s.stop_at_poll = int(stop_at_poll)
    r = _LbaResult()
    chi2 = np.zeros(n_edges, np.float64)
    dpos = np.zeros(n_edges, np.uint8)
    outl = np.zeros(n_edges, np.uint8)
    lvl1 = np.zeros(n_edges, np.uint8)
    r.edge_chi2, r.edge_depth_pos, r.edge_outlier, r.edge_level1 = (a.ctypes.data for a in (chi2, dpos, outl, lvl1))
    L = contracted_lib() if contracted else lib()
    L.orc_set_lba_variant(int(variant))
    try:
        st = L.orc_lba_solve(C.byref(s), C.byref(r))
    finally:
        L.orc_set_lba_variant(0)
    Tout = np.stack([pose_to_Tcw(qt[i]) for i in range(n_poses)])
    return dict(status=st, pose_qt=qt, pose_Tcw=Tout, point_xyz=pts.astype(np.float32), point_xyz64=pts,
                edge_chi2=chi2, edge_depth_pos=dpos, edge_outlier=outl, edge_level1=lvl1,
                lambda_trace=np.array(r.lambda_trace[: r.n_trace]), chi2_trace=np.array(r.chi2_trace[: r.n_trace]),
                iters=(r.iters_done1, r.iters_done2), polls=r.polls, stop_poll=r.stop_poll, trials=r.trials)


class _PoseProblem(C.Structure):
    _fields_ = [("n", C.c_int), ("Xw", C.c_void_p), ("obs", C.c_void_p), ("stereo", C.c_void_p),
                ("inv_sigma2", C.c_void_p), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("bf", C.c_double)]


def pose_optimization(prob: dict):
    """Optimizer::PoseOptimization on a synth_pose_problem()-style dict; float32 boundary like the reference."""
    L = lib()
    L.orc_pose_optimization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    n = int(prob["n"])
    Xw = np.ascontiguousarray(prob["Xw"], np.float32).astype(np.float64)
    obs = np.ascontiguousarray(prob["obs"], np.float32).astype(np.float64)
    st = np.ascontiguousarray(prob["stereo"], np.uint8)
    w = np.ascontiguousarray(prob["inv_sigma2"], np.float32)
    s = _PoseProblem()
    s.n, s.Xw, s.obs, s.stereo, s.inv_sigma2 = n, Xw.ctypes.data, obs.ctypes.data, st.ctypes.data, w.ctypes.data
    s.fx, s.fy, s.cx, s.cy, s.bf = (float(np.float32(prob[k])) for k in ("fx", "fy", "cx", "cy", "bf"))
    qin = pose_from_Tcw(prob["Tcw"])
    qout = np.zeros(7)
    outl = np.zeros(max(n, 1), np.uint8)
    nbad = C.c_int(0)
    ninl = L.orc_pose_optimization(C.byref(s), _p(qin), _p(qout), _p(outl), C.byref(nbad))
    return dict(n_inliers=ninl, n_bad=nbad.value, outlier=outl[:n].copy(), pose_qt=qout, Tcw=pose_to_Tcw(qout))


def se3_exp(upd):
    upd = np.ascontiguousarray(upd, np.float64)
    out = np.zeros(7)
    lib().orc_se3_exp(_p(upd), _p(out))
    return out


def se3_mul(a, b):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    out = np.zeros(7)
    lib().orc_se3_mul(_p(a), _p(b), _p(out))
    return out


def edge_linearize(qt, xyz, obs, stereo, fx, fy, cx, cy, bf):
    qt = np.ascontiguousarray(qt, np.float64)
    xyz = np.ascontiguousarray(xyz, np.float64)
    obs = np.ascontiguousarray(obs, np.float64)
    err, Ji, Jj = np.zeros(3), np.zeros(9), np.zeros(18)
    lib().orc_edge_linearize(_p(qt), _p(xyz), _p(obs), int(stereo), fx, fy, cx, cy, bf, _p(err), _p(Ji), _p(Jj))
    return err, Ji.reshape(3, 3), Jj.reshape(3, 6)

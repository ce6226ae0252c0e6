"""ctypes binding of the C ABI declared in include/aos2.h (lib/libaos2.so, built by csrc/Makefile).

This is plumbing: numpy arrays in, numpy arrays out, every call goes through the extern "C" entry
points a C++/cgo/JNI caller would use.  There is no Python or CPU fallback: a missing library
raises LibraryMissing, a missing GPU surfaces as AosError(AOS2_ERR_NO_DEVICE).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

AOS2_OK = 0
AOS2_ERR_ARG, AOS2_ERR_CAPACITY, AOS2_ERR_TOO_SMALL, AOS2_ERR_HIP, AOS2_ERR_NO_DEVICE = -1, -2, -3, -4, -5
AOS2_ERR_STOPPED = 1

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


class LibraryMissing(RuntimeError):
    pass


class AosError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"aos2 error {code}: {msg}")
        self.code = code


def lib_path():
    return os.environ.get("AOS2_LIB") or os.path.join(_HERE, "lib", "libaos2.so")


def lib():
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise LibraryMissing(f"{p} not found: build it with `make -C {_HERE}/csrc` "
                                 "(__graft_entry__.build()); there is no fallback path")
        L = C.CDLL(p)
        L.aos2_last_error.restype = C.c_char_p
        L.aos2_version.restype = C.c_char_p
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        L.aos2_extractor_create.argtypes = [ci, cf, ci, ci, ci, ci, C.POINTER(vp)]
        L.aos2_extractor_destroy.argtypes = [vp]
        L.aos2_extractor_levels.argtypes = [vp]
        L.aos2_extractor_scale_factor.argtypes = [vp]
        L.aos2_extractor_scale_factor.restype = cf
        for n in ("scale_factors", "inv_scale_factors", "sigma2", "inv_sigma2"):
            f = getattr(L, "aos2_extractor_" + n)
            f.argtypes = [vp]
            f.restype = C.POINTER(cf)
        for n in ("features_per_level", "umax"):
            f = getattr(L, "aos2_extractor_" + n)
            f.argtypes = [vp]
            f.restype = C.POINTER(ci)
        L.aos2_extractor_max_keypoints.argtypes = [vp]
        L.aos2_extractor_max_keypoints_for.argtypes = [vp, ci, ci]
        L.aos2_extractor_extract.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, C.POINTER(ci)]
        L.aos2_extractor_extract_batch.argtypes = [vp, vp, ci, ci, ci, ci, C.c_size_t, vp, vp, ci, vp]
        L.aos2_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
        L.aos2_host_free.argtypes = [vp]
        L.aos2_extractor_extract_batch_device.argtypes = [vp, vp, ci, ci, ci, ci, C.c_size_t, vp, vp, ci, vp]
        L.aos2_extractor_extract_batch_device_async.argtypes = [vp, vp, ci, ci, ci, ci, C.c_size_t, vp, vp, ci, vp]
        L.aos2_extractor_wait.argtypes = [vp]
        L.aos2_extractor_pyramid_level_size.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci)]
        L.aos2_extractor_pyramid_level.argtypes = [vp, ci, ci, ci, vp, ci]
        L.aos2_extractor_debug_candidates.argtypes = [vp, ci, ci, vp, vp, vp, ci, C.POINTER(ci)]
        L.aos2_extractor_last_timing.argtypes = [vp, vp, ci]
        if hasattr(L, "aos2_extractor_set_chunks"):
            L.aos2_extractor_set_chunks.argtypes = [vp, ci]
        L.aos2_extractor_bench_fast.argtypes = [vp, ci, C.POINTER(cf)]
        L.aos2_extractor_bench_describe.argtypes = [vp, ci, C.POINTER(cf)]
        if hasattr(L, "aos2_compute_stereo_matches"):
            L.aos2_compute_stereo_matches.argtypes = [vp, vp, ci, vp, vp, ci, vp, vp, ci, cf, cf, vp, vp]
            L.aos2_compute_stereo_matches_device.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, cf, cf, vp, vp]
            L.aos2_compute_stereo_matches_device_async.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, cf, cf, vp, vp]
            L.aos2_compute_stereo_matches_last_device_ms.restype = cf
            L.aos2_compute_stereo_matches_last_device_ms.argtypes = [vp]
        if hasattr(L, "aos2_vocabulary_create"):
            L.aos2_vocabulary_create.argtypes = [ci, C.POINTER(vp)]
            L.aos2_vocabulary_destroy.argtypes = [vp]
            L.aos2_vocabulary_load_binary.argtypes = [vp, C.c_char_p]
            L.aos2_vocabulary_load_text.argtypes = [vp, C.c_char_p]
            L.aos2_vocabulary_save_binary.argtypes = [vp, C.c_char_p]
            L.aos2_vocabulary_set_nodes.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp]
            for nm in ("k", "levels", "scoring", "weighting", "nodes", "empty"):
                getattr(L, "aos2_vocabulary_" + nm).argtypes = [vp]
            L.aos2_vocabulary_size.argtypes = [vp]
            L.aos2_vocabulary_size.restype = C.c_uint
            L.aos2_vocabulary_transform.argtypes = [vp, vp, ci, ci, vp, vp, C.POINTER(ci), vp, vp, vp, C.POINTER(ci), vp, vp]
            L.aos2_vocabulary_transform_device.argtypes = [vp, ci, vp, vp, ci, ci] + [vp] * 9
            L.aos2_vocabulary_last_device_ms.restype = cf
            L.aos2_vocabulary_last_device_ms.argtypes = [vp]
            L.aos2_vocabulary_score.argtypes = [vp, vp, vp, ci, vp, vp, ci, C.POINTER(C.c_double)]
        L.aos2_debug_octree_host.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, ci]
        L.aos2_debug_sincos_host.argtypes = [cf, C.POINTER(cf), C.POINTER(cf)]
        L.aos2_debug_sincos_device.argtypes = [vp, ci, vp, vp, ci]
        L.aos2_debug_pose_blocks_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci]
        if hasattr(L, "aos2_matcher_create"):
            L.aos2_matcher_create.argtypes = [cf, ci, ci, C.POINTER(vp)]
            L.aos2_matcher_destroy.argtypes = [vp]
            L.aos2_descriptor_distance.argtypes = [vp, vp]
            if hasattr(L, "aos2_matcher_last_device_ms"):
                L.aos2_matcher_last_device_ms.argtypes = [vp]
                L.aos2_matcher_last_device_ms.restype = cf
            L.aos2_matcher_hamming_best2.argtypes = [vp, vp, ci, vp, ci, vp, vp, vp]
            L.aos2_matcher_hamming_best2_device.argtypes = [vp, vp, ci, vp, ci, vp, vp, vp, ci, C.POINTER(cf)]
            L.aos2_matcher_search_by_bow.argtypes = [vp, vp, ci, vp, vp]
            L.aos2_matcher_search_by_bow_device.argtypes = [vp, vp, ci, vp, vp]
            L.aos2_matcher_search_by_bow_frames.argtypes = [vp, vp, vp, vp]
            L.aos2_vocabulary_stream.argtypes = [vp]
            L.aos2_vocabulary_stream.restype = vp
            if hasattr(L, "aos2_matcher_search_by_bow_kf"):
                L.aos2_matcher_search_by_bow_kf.argtypes = [vp, vp, ci, vp, vp]
                L.aos2_matcher_search_for_triangulation.argtypes = [vp, vp, ci, ci, vp, vp]
                L.aos2_compute_distinctive_descriptors.argtypes = [vp, ci, vp, vp, vp]
            if hasattr(L, "aos2_matcher_fuse"):
                L.aos2_matcher_fuse.argtypes = [vp, vp, vp, ci, vp, vp, vp]
                L.aos2_matcher_search_by_projection_kf.argtypes = [vp, vp, vp, vp, vp]
                L.aos2_matcher_search_by_sim3.argtypes = [vp, vp, vp, vp, vp, vp, vp]
                L.aos2_matcher_search_by_projection_reloc.argtypes = [vp, vp, vp, ci, vp, vp]
            if hasattr(L, "aos2_frame_assign_features_to_grid"):
                L.aos2_frame_assign_features_to_grid.argtypes = [vp, ci, vp, vp, cf, cf, cf, cf, vp, vp, C.POINTER(C.c_int32)]
                L.aos2_frame_stereo_from_rgbd.argtypes = [vp, ci, vp, vp, vp, vp, ci, ci, ci, cf, vp, vp]
            if hasattr(L, "aos2_frame_is_in_frustum"):
                L.aos2_frame_is_in_frustum.argtypes = [vp, vp, cf, cf, cf, cf, ci, cf, vp, vp, vp, vp, vp, vp]
            if hasattr(L, "aos2_matcher_search_for_initialization"):
                L.aos2_matcher_search_for_initialization.argtypes = [vp, vp, ci, vp, vp, vp, vp, ci, vp, vp]
            if hasattr(L, "aos2_matcher_search_by_projection_batch"):
                L.aos2_matcher_search_by_projection_batch.argtypes = [vp, vp, vp, ci, cf, vp, vp]
            L.aos2_matcher_search_by_projection.argtypes = [vp, vp, vp, cf, vp, vp]
            L.aos2_matcher_search_by_projection_last.argtypes = [vp, vp, vp, cf, ci, vp, vp]
        if hasattr(L, "aos2_lba_create"):
            L.aos2_lba_create.argtypes = [ci, C.POINTER(vp)]
            L.aos2_lba_destroy.argtypes = [vp]
            L.aos2_lba_solve.argtypes = [vp, vp, vp]
            L.aos2_lba_solve_batch.argtypes = [vp, vp, vp, ci]
            L.aos2_lba_debug_stop_at_poll.argtypes = [vp, ci]
            L.aos2_lba_set_host_threads.argtypes = [vp, ci]
            L.aos2_lba_set_window_groups.argtypes = [vp, ci]
            L.aos2_lba_last_program.argtypes = [vp, vp, vp]
            L.aos2_lba_debug_host_phase.argtypes = [vp, ci, ci, vp, vp]
            if hasattr(L, "aos2_pose_optimization"):
                L.aos2_pose_optimization.argtypes = [vp, vp, vp, ci]
                L.aos2_pose_optimization_last_device_ms.argtypes = [vp]
                L.aos2_pose_optimization_last_device_ms.restype = cf
        if hasattr(L, "aos2_frames_create"):
            L.aos2_frames_create.argtypes = [ci, ci, ci, C.POINTER(vp)]
            L.aos2_frames_destroy.argtypes = [vp]
            L.aos2_frames_stream.argtypes = [vp]
            L.aos2_frames_stream.restype = vp
            L.aos2_frames_wait.argtypes = [vp]
            L.aos2_frames_wait_for_stream.argtypes = [vp, vp]
            L.aos2_frames_device_ptr.argtypes = [vp, ci]
            L.aos2_frames_search_for_triangulation.argtypes = [vp, vp, vp, ci, ci, vp, vp]
            L.aos2_frames_fuse.argtypes = [vp, vp, ci, ci, vp, vp, cf, vp, vp]
            L.aos2_frames_device_ptr.restype = vp
            L.aos2_frames_build.argtypes = [vp, vp, ci, vp, vp, vp, ci, ci, ci, vp, ci, C.c_size_t, cf, cf, cf, cf, cf]
            L.aos2_frames_build_stereo.argtypes = [vp, vp, ci, vp, vp, vp, ci, ci, ci, vp, vp, cf, cf, cf, cf, cf]
            L.aos2_frames_set_pose.argtypes = [vp, vp]
            L.aos2_frames_set_distortion.argtypes = [vp, cf, cf, cf, cf, cf]
            L.aos2_frame_image_bounds.argtypes = [ci, ci, cf, cf, cf, cf, vp, vp]
            L.aos2_frames_set_map_points.argtypes = [vp, vp, vp, vp]
            L.aos2_frames_get.argtypes = [vp, ci, vp, C.c_size_t]
            L.aos2_frames_search_by_projection_last.argtypes = [vp, vp, vp, cf, ci, ci, vp]
            L.aos2_frames_pose_optimization.argtypes = [vp, vp, vp]
            L.aos2_frames_discard_outliers.argtypes = [vp]
            L.aos2_frames_search_local_points.argtypes = [vp, vp, vp, ci, cf, cf, vp]
            L.aos2_extractor_stream_wait.argtypes = [vp, vp]
            L.aos2_extractor_wait_for_stream.argtypes = [vp, vp]
            L.aos2_lba_last_window_slots.argtypes = [vp, C.POINTER(C.c_int64)]
            L.aos2_extractor_pack_slots.argtypes = [vp, ci, vp, vp, vp, ci, vp, C.c_size_t, vp]
            L.aos2_capture_begin.argtypes = [vp]
            L.aos2_capture_end.argtypes = [vp, C.POINTER(vp)]
            L.aos2_graph_launch.argtypes = [vp, vp]
            L.aos2_graph_nodes.argtypes = [vp]
            L.aos2_graph_destroy.argtypes = [vp]
            L.aos2_graph_destroy.restype = None
        _LIB = _RecLib(L)
    return _LIB


# ---- recording (bench harness): the calls of a job captured instead of executed, for the native step runner (csrc/host_runner.cpp)
class _Call(C.Structure):
    _fields_ = [("fn", C.c_void_p), ("n_int", C.c_int32), ("n_fp", C.c_int32), ("iargs", C.c_int64 * 24), ("fargs", C.c_uint64 * 8)]


# the functions a recorded job may consist of: work that is enqueued / solved / waited for.  Everything else (getters, stream handles,
# setters) executes at once even while a recording is open
_RECORDABLE = {
    "aos2_frames_set_pose", "aos2_extractor_extract_batch_device_async", "aos2_frames_build", "aos2_frames_build_stereo",
    "aos2_compute_stereo_matches_device_async", "aos2_frames_search_by_projection_last", "aos2_frames_pose_optimization",
    "aos2_frames_discard_outliers", "aos2_frames_search_local_points", "aos2_frames_wait", "aos2_extractor_wait",
    "aos2_extractor_stream_wait", "aos2_extractor_wait_for_stream", "aos2_vocabulary_transform_device", "aos2_matcher_search_by_bow_frames",
    "aos2_frames_search_for_triangulation", "aos2_frames_fuse", "aos2_lba_solve_batch", "aos2_extractor_pack_slots",
    "hipMemcpyAsync", "hipStreamSynchronize", "hipStreamWaitEvent", "hipEventRecord", "hipMemcpy",
}
_REC = None   # the open recording (a list), or None


class recording:
    """with capi.recording() as calls: ...   -- the recordable C calls made inside are appended to `calls` (not executed; they
    "return" AOS2_OK); calls.keep holds every argument object alive; calls.array() is the aos2_call_t array for the runner"""

    class _List(list):
        def __init__(self):
            super().__init__()
            self.keep = []
            self.names = []   # function name per call (diagnostics)

        def array(self):
            a = (_Call * max(1, len(self)))()
            for i, c in enumerate(self):
                a[i] = c
            return a

    def __enter__(self):
        global _REC
        assert _REC is None, "recordings do not nest"
        _REC = recording._List()
        return _REC

    def __exit__(self, *exc):
        global _REC
        _REC = None
        return False


def _record(fn, name, args):
    import struct
    at = fn.argtypes
    if at is None or len(at) != len(args):
        raise TypeError(f"recording {name}: the function needs argtypes for every argument")
    c = _Call()
    c.fn = C.cast(fn, C.c_void_p).value
    ni = nf = 0
    for a, t in zip(args, at):
        if t is C.c_float or t is C.c_double:
            v = float(a.value if isinstance(a, C._SimpleCData) else a)
            bits = struct.unpack("<I", struct.pack("<f", v))[0] if t is C.c_float else struct.unpack("<Q", struct.pack("<d", v))[0]
            if nf >= 8:
                raise TypeError(f"recording {name}: more than 8 floating-point arguments")
            c.fargs[nf] = bits
            nf += 1
            continue
        if a is None:
            v = 0
        elif isinstance(a, (int, np.integer)):
            v = int(a)
        elif isinstance(a, C._SimpleCData):
            v = a.value or 0
        else:
            v = C.cast(a, C.c_void_p).value or 0   # pointers, arrays, byref()
        if ni >= 24:
            raise TypeError(f"recording {name}: more than 24 integer arguments")
        if v >= 1 << 63:
            v -= 1 << 64
        c.iargs[ni] = v
        ni += 1
    c.n_int, c.n_fp = ni, nf
    _REC.append(c)
    _REC.keep.append(args)
    _REC.names.append(name)
    return AOS2_OK


class _RecLib:
    """the loaded library; a recordable function called while a recording is open is captured instead of executed"""

    def __init__(self, L):
        object.__setattr__(self, "_L", L)
        object.__setattr__(self, "_w", {})
        object.__setattr__(self, "_names", frozenset(_RECORDABLE))   # (kept here: module globals are gone when handles close at exit)

    def __getattr__(self, name):
        w = self._w.get(name)
        if w is None:
            fn = getattr(self._L, name)
            if name in self._names:
                def w(*args, _fn=fn, _name=name):
                    if _REC is None:
                        return _fn(*args)
                    return _record(_fn, _name, args)
                w.argtypes_of = fn
            else:
                w = fn
            self._w[name] = w
        return w


class Graph:
    """include/aos2.h "Replay of a fixed call sequence": the calls made inside `with Graph.capture(stream) as g:` are recorded
    (nothing runs); g.launch() enqueues them all with one call"""

    def __init__(self, stream):
        self.L, self.stream, self.h = lib(), stream, None

    @classmethod
    def capture(cls, stream):
        return cls(stream)

    def __enter__(self):
        _check(self.L.aos2_capture_begin(self.stream))
        return self

    def __exit__(self, et, ev, tb):
        h = C.c_void_p()
        st = self.L.aos2_capture_end(self.stream, C.byref(h))
        if et is None:
            _check(st)
            self.h = h
        elif st == AOS2_OK and h:
            self.L.aos2_graph_destroy(h)   # (the body failed: what was recorded until then is dropped)
        return False

    def nodes(self):
        return int(self.L.aos2_graph_nodes(self.h))

    def launch(self, stream=None):
        _check(self.L.aos2_graph_launch(self.h, self.stream if stream is None else stream))

    def close(self):
        if self.h:
            self.L.aos2_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_HIP = None


def hip_runtime():
    """libamdhip64 (the runtime the library itself links) for the harness's own copies: hipMemcpyAsync & co, recordable"""
    global _HIP
    if _HIP is None:
        H = C.CDLL("libamdhip64.so")
        vp = C.c_void_p
        H.hipMemcpyAsync.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
        H.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
        H.hipStreamSynchronize.argtypes = [vp]
        H.hipStreamWaitEvent.argtypes = [vp, vp, C.c_uint]
        H.hipEventRecord.argtypes = [vp, vp]
        H.hipStreamCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]
        H.hipEventCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]
        H.hipHostMalloc.argtypes = [C.POINTER(vp), C.c_size_t, C.c_uint]
        _HIP = _RecLib(H)
    return _HIP


HIP_D2H, HIP_H2D = 2, 1


class Runner:
    """csrc/host_runner.cpp: the step schedule of bench.py on native threads over recorded call lists"""
    PIPE_WAIT, PIPE_STEP, KF_JOB, LBA_JOB = 0, 1, 2, 3

    def __init__(self, n_pipes, n_lba):
        p = os.path.join(os.path.dirname(lib_path()), "libaos2_runner.so")
        if not os.path.exists(p):
            raise LibraryMissing(f"{p} not found: build it with `make -C {_HERE}/csrc`")
        R = C.CDLL(p)
        vp = C.c_void_p
        R.aos2_runner_create.restype = vp
        R.aos2_runner_create.argtypes = [C.c_int, C.c_int]
        R.aos2_runner_destroy.argtypes = [vp]
        R.aos2_runner_set_list.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
        R.aos2_runner_step.argtypes = [vp, C.c_int]
        R.aos2_runner_run.argtypes = [vp, C.c_int, C.c_int]
        R.aos2_runner_sync.argtypes = [vp]
        R.aos2_runner_error.argtypes = [vp]
        R.aos2_runner_error.restype = C.c_char_p
        R.aos2_runner_reset_stats.argtypes = [vp]
        R.aos2_runner_stats.argtypes = [vp, C.c_int, vp, C.c_int]
        R.aos2_runner_set_lba_every.argtypes = [vp, C.c_int]
        R.aos2_runner_last_lba.argtypes = [vp]
        self.R = R
        self.h = R.aos2_runner_create(int(n_pipes), int(n_lba))
        if not self.h:
            raise ValueError("aos2_runner_create")
        self._keep = {}

    def __del__(self):
        if getattr(self, "h", None):
            self.R.aos2_runner_destroy(self.h)
            self.h = None

    def set_list(self, kind, index, calls):
        """calls: a recording (or None / empty: the job is skipped)"""
        n = len(calls) if calls else 0
        arr = calls.array() if n else None
        if self.R.aos2_runner_set_list(self.h, kind, index, C.cast(arr, C.c_void_p) if n else None, n) != 0:
            raise ValueError("aos2_runner_set_list")
        self._keep[(kind, index)] = (calls, arr)

    def _st(self, st):
        if st != 0:
            raise AosError(st, self.R.aos2_runner_error(self.h).decode() + ": " + lib().aos2_last_error().decode(errors="replace"))

    def set_lba_every(self, n):
        """a LocalBA job every n steps (its list then solves the windows of n steps in one call)"""
        if self.R.aos2_runner_set_lba_every(self.h, int(n)) != 0:
            raise ValueError("aos2_runner_set_lba_every")

    def last_lba(self):
        return int(self.R.aos2_runner_last_lba(self.h))

    def step(self, s):
        self._st(self.R.aos2_runner_step(self.h, int(s)))

    def run(self, s0, n):
        self._st(self.R.aos2_runner_run(self.h, int(s0), int(n)))

    def sync(self):
        self._st(self.R.aos2_runner_sync(self.h))

    def reset_stats(self):
        self.R.aos2_runner_reset_stats(self.h)

    def stats(self, which):
        n = self.R.aos2_runner_stats(self.h, which, None, 0)
        a = np.zeros(max(n, 1), np.float64)
        self.R.aos2_runner_stats(self.h, which, _p(a), n)
        return a[:n]


def _check(st, ok=(AOS2_OK,)):
    if st not in ok:
        raise AosError(st, lib().aos2_last_error().decode(errors="replace"))
    return st


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class _HostBlock:
    """page-locked host allocation (aos2_host_alloc), released with the last array that views it"""

    def __init__(self, nbytes):
        self.L = lib()
        p = C.c_void_p()
        _check(self.L.aos2_host_alloc(C.byref(p), max(int(nbytes), 1)))
        self.p = p.value

    def __del__(self):
        if getattr(self, "p", None) and C is not None:   # (at interpreter shutdown the module globals may already be gone)
            self.L.aos2_host_free(C.c_void_p(self.p))
            self.p = None


def host_empty(shape, dtype=np.uint8):
    """uninitialised numpy array in page-locked host memory (for Extractor.extract_batch inputs / outputs)"""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    blk = _HostBlock(n)
    buf = (C.c_uint8 * max(n, 1)).from_address(blk.p)
    buf._block = blk   # the ctypes view keeps the allocation alive; numpy keeps the view alive (.base)
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


def device_count():
    return lib().aos2_device_count()


def device_local_cpus(device=0):
    """set of host CPUs on the device's NUMA node (aos2_device_local_cpus; empty when the platform does not say)"""
    L = lib()
    L.aos2_device_local_cpus.argtypes = [C.c_int, C.c_char_p, C.c_int]
    buf = C.create_string_buffer(8192)
    _check(L.aos2_device_local_cpus(int(device), buf, len(buf)))
    cpus = set()
    for part in buf.value.decode().split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def bind_to_device_node(device=0):
    """sched_setaffinity of the calling thread (and the threads it creates from now on) to the device's local CPUs, intersected with
    the CPUs the process may use; returns the CPU count bound to, 0 when nothing was changed"""
    import os
    try:
        cpus = device_local_cpus(device) & os.sched_getaffinity(0)
    except (AosError, OSError):
        return 0
    if not cpus:
        return 0
    os.sched_setaffinity(0, cpus)
    return len(cpus)


class Extractor:
    """Mirror of ORBextractor (include/ORBextractor.h:45-111): ctor args, operator(), getters."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, device=0):
        self.L = lib()
        h = C.c_void_p()
        _check(self.L.aos2_extractor_create(nfeatures, scale_factor, nlevels, ini_th, min_th, device, C.byref(h)))
        self.h = h
        self.nlevels = nlevels
        self.nfeatures = nfeatures

    def close(self):
        if getattr(self, "h", None):
            self.L.aos2_extractor_destroy(self.h)
            self.h = None

    __del__ = close

    def _arr(self, name, n, dt):
        ptr = getattr(self.L, "aos2_extractor_" + name)(self.h)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt).copy()

    # GetLevels / GetScaleFactor(s) / ... (include/ORBextractor.h:58-83)
    def GetLevels(self):
        return self.L.aos2_extractor_levels(self.h)

    def GetScaleFactor(self):
        return self.L.aos2_extractor_scale_factor(self.h)

    def GetScaleFactors(self):
        return self._arr("scale_factors", self.nlevels, np.float32)

    def GetInverseScaleFactors(self):
        return self._arr("inv_scale_factors", self.nlevels, np.float32)

    def GetScaleSigmaSquares(self):
        return self._arr("sigma2", self.nlevels, np.float32)

    def GetInverseScaleSigmaSquares(self):
        return self._arr("inv_sigma2", self.nlevels, np.float32)

    @property
    def features_per_level(self):
        return self._arr("features_per_level", self.nlevels, np.int32)

    @property
    def umax(self):
        return self._arr("umax", 16, np.int32)

    @property
    def max_keypoints(self):
        return self.L.aos2_extractor_max_keypoints(self.h)

    def max_keypoints_for(self, w, h):
        """capacity that always suffices for w x h images (wide images can exceed nfeatures + 3 per level)"""
        return self.L.aos2_extractor_max_keypoints_for(self.h, int(w), int(h))

    def __call__(self, image, mask=None):
        """operator()(image, mask, keypoints, descriptors): returns (keypoints[KP_DTYPE], desc[n,32])."""
        if image is None or image.size == 0:
            n = C.c_int(-1)
            _check(self.L.aos2_extractor_extract(self.h, None, 0, 0, 0, None, None, 0, C.byref(n)))
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise AssertionError("image.type() == CV_8UC1")  # src/ORBextractor.cc:1050
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = self.max_keypoints_for(w, h)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        _check(self.L.aos2_extractor_extract(self.h, _p(image), w, h, image.strides[0], _p(kps), _p(desc), cap, C.byref(n)))
        return kps[: n.value].copy(), desc[: n.value].copy()

    def extract_batch(self, images):
        """host images [B, h, w] (any host memory; host_empty() arrays are uploaded by DMA) -> [(kps, desc)] * B"""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        B, h, w = images.shape
        cap = self.max_keypoints_for(w, h)
        if getattr(self, "_out_key", None) != (B, cap):   # page-locked result buffers, reused between calls
            self._out_key, self._out = None, None
            self._out = (host_empty((B, cap), KP_DTYPE), host_empty((B, cap, 32), np.uint8))
            self._out_key = (B, cap)
        kps, desc = self._out
        n = np.zeros(B, np.int32)
        _check(self.L.aos2_extractor_extract_batch(self.h, _p(images), B, w, h, w, w * h, _p(kps), _p(desc), cap, _p(n)))
        return [(kps[b, : n[b]].copy(), desc[b, : n[b]].copy()) for b in range(B)]

    def extract_batch_device(self, d_imgs, batch, w, h, stride, image_stride, d_kps, d_desc, cap, d_nout):
        """raw device pointers (ints)"""
        _check(self.L.aos2_extractor_extract_batch_device(self.h, C.c_void_p(d_imgs), batch, w, h, stride, image_stride,
                                                          C.c_void_p(d_kps), C.c_void_p(d_desc), cap, C.c_void_p(d_nout)))

    def extract_batch_device_async(self, d_imgs, batch, w, h, stride, image_stride, d_kps, d_desc, cap, d_nout):
        """enqueue only (raw device pointers); results and errors are complete after wait()"""
        _check(self.L.aos2_extractor_extract_batch_device_async(self.h, C.c_void_p(d_imgs), batch, w, h, stride, image_stride,
                                                                C.c_void_p(d_kps), C.c_void_p(d_desc), cap, C.c_void_p(d_nout)))

    def pack_slots(self, batch, d_kps, d_desc, d_n, cap, d_slots, slot_bytes, stream=None):
        """aos2_extractor_pack_slots: device outputs -> fixed-size per-frame slots (sharding.slot_bytes layout)"""
        _check(self.L.aos2_extractor_pack_slots(self.h, batch, d_kps, d_desc, d_n, cap, d_slots, slot_bytes, stream))

    def stream_wait(self, stream):
        _check(self.L.aos2_extractor_stream_wait(self.h, stream))

    def wait_for_stream(self, stream):
        """the batches enqueued from now on run behind what `stream` holds so far (device-side)"""
        _check(self.L.aos2_extractor_wait_for_stream(self.h, stream))

    def wait(self):
        """complete every batch enqueued with extract_batch_device_async"""
        _check(self.L.aos2_extractor_wait(self.h))

    def pyramid_level(self, level, image=0, border=0):
        """mvImagePyramid[level] (include/ORBextractor.h:85)"""
        w, h = C.c_int(), C.c_int()
        _check(self.L.aos2_extractor_pyramid_level_size(self.h, level, C.byref(w), C.byref(h)))
        out = np.zeros((h.value + 2 * border, w.value + 2 * border), np.uint8)
        _check(self.L.aos2_extractor_pyramid_level(self.h, image, level, border, _p(out), out.strides[0]))
        return out

    def debug_candidates(self, level, image=0):
        n = C.c_int(0)
        _check(self.L.aos2_extractor_debug_candidates(self.h, image, level, None, None, None, 0, C.byref(n)))
        xs = np.zeros(max(n.value, 1), np.int16)
        ys = np.zeros(max(n.value, 1), np.int16)
        sc = np.zeros(max(n.value, 1), np.uint8)
        _check(self.L.aos2_extractor_debug_candidates(self.h, image, level, _p(xs), _p(ys), _p(sc), len(xs), C.byref(n)))
        return xs[: n.value], ys[: n.value], sc[: n.value]

    def set_chunks(self, n):
        _check(self.L.aos2_extractor_set_chunks(self.h, n))

    def last_timing(self):
        t = np.zeros(8, np.float32)
        _check(self.L.aos2_extractor_last_timing(self.h, _p(t), 8))
        return dict(pyramid=float(t[0]), fast=float(t[1]), compact=float(t[2]), octree=float(t[3]),
                    describe=float(t[4]), total_wall=float(t[5]), chunks=int(t[6]))

    def bench_fast(self, iters=20):
        ms = C.c_float(0)
        _check(self.L.aos2_extractor_bench_fast(self.h, iters, C.byref(ms)))
        return ms.value

    def bench_describe(self, iters=20):
        ms = C.c_float(0)
        _check(self.L.aos2_extractor_bench_describe(self.h, iters, C.byref(ms)))
        return ms.value


class Vocabulary:
    """ORBVocabulary (include/ORBVocabulary.h:31-32): loaders, transform(), score()."""

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        _check(self.L.aos2_vocabulary_create(device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.aos2_vocabulary_destroy(self.h)
            self.h = None

    __del__ = close

    def loadFromBinaryFile(self, path):
        return self.L.aos2_vocabulary_load_binary(self.h, str(path).encode()) == AOS2_OK

    def loadFromTextFile(self, path):
        return self.L.aos2_vocabulary_load_text(self.h, str(path).encode()) == AOS2_OK

    def saveToBinaryFile(self, path):
        _check(self.L.aos2_vocabulary_save_binary(self.h, str(path).encode()))

    def set_nodes(self, k, L, scoring, weighting, parent, desc, weight, is_leaf):
        parent = np.ascontiguousarray(parent, np.int32)
        desc = np.ascontiguousarray(desc, np.uint8)
        weight = np.ascontiguousarray(weight, np.float64)
        is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        _check(self.L.aos2_vocabulary_set_nodes(self.h, k, L, scoring, weighting, len(parent), _p(parent), _p(desc),
                                                _p(weight), _p(is_leaf)))

    def info(self):
        g = lambda n: getattr(self.L, "aos2_vocabulary_" + n)(self.h)  # noqa: E731
        return dict(k=g("k"), L=g("levels"), scoring=g("scoring"), weighting=g("weighting"), nodes=g("nodes"),
                    words=int(g("size")))

    def empty(self):
        return bool(self.L.aos2_vocabulary_empty(self.h))

    def transform(self, desc, levelsup=4):
        """-> dict(bow_word, bow_value, fv_node, fv_off, fv_idx, word_of, node_of)"""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        m = max(n, 1)
        bw, bv = np.zeros(m, np.uint32), np.zeros(m, np.float64)
        fn, fo, fi = np.zeros(m, np.int32), np.zeros(n + 2, np.int32), np.zeros(m, np.int32)
        wo, no = np.zeros(m, np.uint32), np.zeros(m, np.uint32)
        nb, nf = C.c_int(0), C.c_int(0)
        _check(self.L.aos2_vocabulary_transform(self.h, _p(desc), n, levelsup, _p(bw), _p(bv), C.byref(nb), _p(fn), _p(fo),
                                                _p(fi), C.byref(nf), _p(wo), _p(no)))
        nb, nf = nb.value, nf.value
        return dict(bow_word=bw[:nb].copy(), bow_value=bv[:nb].copy(), fv_node=fn[:nf].copy(), fv_off=fo[: nf + 1].copy(),
                    fv_idx=fi[: fo[nf]].copy(), word_of=wo[:n].copy(), node_of=no[:n].copy())

    def transform_device(self, batch, d_desc, d_n, cap, levelsup, d_bw, d_bv, d_nb, d_fn, d_fo, d_fi, d_nf, d_wo=0, d_no=0):
        """raw device pointers (ints); returns the device time in ms"""
        V = C.c_void_p
        _check(self.L.aos2_vocabulary_transform_device(self.h, batch, V(d_desc), V(d_n), cap, levelsup, V(d_bw), V(d_bv),
                                                       V(d_nb), V(d_fn), V(d_fo), V(d_fi), V(d_nf), V(d_wo or None),
                                                       V(d_no or None)))
        return float(self.L.aos2_vocabulary_last_device_ms(self.h))

    def stream(self):
        return self.L.aos2_vocabulary_stream(self.h)

    def score(self, a, b):
        w1, v1 = np.ascontiguousarray(a["bow_word"], np.uint32), np.ascontiguousarray(a["bow_value"], np.float64)
        w2, v2 = np.ascontiguousarray(b["bow_word"], np.uint32), np.ascontiguousarray(b["bow_value"], np.float64)
        s = C.c_double(0)
        _check(self.L.aos2_vocabulary_score(self.h, _p(w1), _p(v1), len(w1), _p(w2), _p(v2), len(w2), C.byref(s)))
        return s.value


def ComputeStereoMatches(left: Extractor, right: Extractor, kps_l, desc_l, kps_r, desc_r, mb, mbf, image=0):
    """Frame::ComputeStereoMatches (src/Frame.cc:495-669) on the pyramids `left` / `right` hold from their
    last extract.  Returns (mvuRight, mvDepth) float32 arrays, -1 = no match."""
    kps_l = np.ascontiguousarray(kps_l, KP_DTYPE)
    kps_r = np.ascontiguousarray(kps_r, KP_DTYPE)
    desc_l = np.ascontiguousarray(desc_l, np.uint8)
    desc_r = np.ascontiguousarray(desc_r, np.uint8)
    ur = np.full(len(kps_l), -1.0, np.float32)
    dp = np.full(len(kps_l), -1.0, np.float32)
    _check(left.L.aos2_compute_stereo_matches(left.h, right.h, image, _p(kps_l), _p(desc_l), len(kps_l), _p(kps_r),
                                              _p(desc_r), len(kps_r), mb, mbf, _p(ur), _p(dp)))
    return ur, dp


def compute_stereo_matches_device(left: Extractor, right: Extractor, batch, d_kpl, d_dl, d_nl, d_kpr, d_dr, d_nr, cap,
                                  mb, mbf, d_ur, d_depth):
    """raw device pointers (ints); arrays as written by extract_batch_device for the two eyes"""
    V = C.c_void_p
    _check(left.L.aos2_compute_stereo_matches_device(left.h, right.h, batch, V(d_kpl), V(d_dl), V(d_nl), V(d_kpr),
                                                     V(d_dr), V(d_nr), cap, mb, mbf, V(d_ur), V(d_depth)))
    return float(left.L.aos2_compute_stereo_matches_last_device_ms(left.h))


def compute_stereo_matches_device_async(left: Extractor, right: Extractor, batch, d_kpl, d_dl, d_nl, d_kpr, d_dr, d_nr, cap,
                                        mb, mbf, d_ur, d_depth):
    """the same, enqueued behind both extractors' batches in flight (no host wait)"""
    V = C.c_void_p
    _check(left.L.aos2_compute_stereo_matches_device_async(left.h, right.h, batch, V(d_kpl), V(d_dl), V(d_nl), V(d_kpr),
                                                           V(d_dr), V(d_nr), cap, mb, mbf, V(d_ur), V(d_depth)))


def debug_octree_host(xs, ys, score, minX, maxX, minY, maxY, N):
    xs = np.ascontiguousarray(xs, np.int16)
    ys = np.ascontiguousarray(ys, np.int16)
    score = np.ascontiguousarray(score, np.uint8)
    out = np.zeros(max(len(xs), 1), np.int32)
    k = lib().aos2_debug_octree_host(_p(xs), _p(ys), _p(score), len(xs), minX, maxX, minY, maxY, N, _p(out), len(out))
    if k < 0:
        raise AosError(k, "octree scratch exhausted")
    return out[:k].copy()


def debug_sincos_host(angle_rad):
    s, c = C.c_float(), C.c_float()
    lib().aos2_debug_sincos_host(float(angle_rad), C.byref(s), C.byref(c))
    return s.value, c.value


def debug_sincos_device(angles, device=0):
    a = np.ascontiguousarray(angles, np.float32)
    s = np.zeros_like(a)
    c = np.zeros_like(a)
    _check(lib().aos2_debug_sincos_device(_p(a), len(a), _p(s), _p(c), device))
    return s, c


def debug_pose_blocks_device(upd, T, Hb, lam, x0, device=0):
    """test tap: the pose solver's exp-map update and 6x6 solve on the device, n independent cases"""
    upd = np.ascontiguousarray(upd, np.float64); T = np.ascontiguousarray(T, np.float64)
    Hb = np.ascontiguousarray(Hb, np.float64); lam = np.ascontiguousarray(lam, np.float64)
    x = np.ascontiguousarray(x0, np.float64).copy()
    n = len(lam)
    To = np.zeros((n, 7)); ok = np.zeros(n, np.uint8)
    _check(lib().aos2_debug_pose_blocks_device(_p(upd), _p(T), _p(To), _p(Hb), _p(lam), _p(x), _p(ok), n, device))
    return To, x, ok


# ------------------------------------------------------------------------------------------ matcher
class _BowPair(C.Structure):
    _fields_ = [("n_kf", C.c_int32), ("n_f", C.c_int32), ("desc_kf", C.c_void_p), ("desc_f", C.c_void_p),
                ("kf_has_mp", C.c_void_p), ("angle_kf", C.c_void_p), ("angle_f", C.c_void_p),
                ("n_nodes_kf", C.c_int32), ("n_nodes_f", C.c_int32),
                ("node_id_kf", C.c_void_p), ("node_off_kf", C.c_void_p), ("node_idx_kf", C.c_void_p),
                ("node_id_f", C.c_void_p), ("node_off_f", C.c_void_p), ("node_idx_f", C.c_void_p)]


class _BowFrames(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("cap", C.c_int32)] + [(k, C.c_void_p) for k in (
        "d_desc_kf", "d_kps_kf", "d_n_kf", "d_desc_f", "d_kps_f", "d_n_f", "kf_has_mp", "d_kf_fv_node", "d_kf_fv_off", "d_kf_fv_idx",
        "d_kf_n_fv", "d_f_fv_node", "d_f_fv_off", "d_f_fv_idx", "d_f_n_fv")]


class _BowKfPair(C.Structure):
    _fields_ = [("n1", C.c_int32), ("n2", C.c_int32), ("desc1", C.c_void_p), ("desc2", C.c_void_p),
                ("has_mp1", C.c_void_p), ("has_mp2", C.c_void_p), ("angle1", C.c_void_p), ("angle2", C.c_void_p),
                ("n_nodes1", C.c_int32), ("n_nodes2", C.c_int32),
                ("node_id1", C.c_void_p), ("node_off1", C.c_void_p), ("node_idx1", C.c_void_p),
                ("node_id2", C.c_void_p), ("node_off2", C.c_void_p), ("node_idx2", C.c_void_p)]


class _TriangPair(C.Structure):
    _fields_ = [("n1", C.c_int32), ("n2", C.c_int32), ("desc1", C.c_void_p), ("desc2", C.c_void_p),
                ("has_mp1", C.c_void_p), ("has_mp2", C.c_void_p),
                ("x1", C.c_void_p), ("y1", C.c_void_p), ("angle1", C.c_void_p), ("u_right1", C.c_void_p),
                ("x2", C.c_void_p), ("y2", C.c_void_p), ("angle2", C.c_void_p), ("u_right2", C.c_void_p),
                ("octave2", C.c_void_p), ("scale_factors2", C.c_void_p), ("level_sigma2_2", C.c_void_p),
                ("n_levels2", C.c_int32), ("F12", C.c_float * 9), ("ex", C.c_float), ("ey", C.c_float),
                ("n_nodes1", C.c_int32), ("n_nodes2", C.c_int32),
                ("node_id1", C.c_void_p), ("node_off1", C.c_void_p), ("node_idx1", C.c_void_p),
                ("node_id2", C.c_void_p), ("node_off2", C.c_void_p), ("node_idx2", C.c_void_p)]


class _FrameView(C.Structure):
    _fields_ = [("n_f", C.c_int32), ("desc_f", C.c_void_p), ("kp_x", C.c_void_p), ("kp_y", C.c_void_p),
                ("kp_octave", C.c_void_p), ("kp_angle", C.c_void_p), ("u_right", C.c_void_p),
                ("scale_factors", C.c_void_p), ("n_levels", C.c_int32),
                ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("grid_w_inv", C.c_float), ("grid_h_inv", C.c_float),
                ("grid_off", C.c_void_p), ("grid_idx", C.c_void_p), ("f_mp_state", C.c_void_p)]


class _ProjMp(C.Structure):
    _fields_ = [("n_mp", C.c_int32), ("track_in_view", C.c_void_p), ("pred_level", C.c_void_p),
                ("view_cos", C.c_void_p), ("proj_x", C.c_void_p), ("proj_y", C.c_void_p),
                ("proj_xr", C.c_void_p), ("desc", C.c_void_p), ("has_obs", C.c_void_p)]


class _ProjLast(C.Structure):
    _fields_ = [("n_last", C.c_int32), ("last_valid", C.c_void_p), ("world_pos", C.c_void_p),
                ("desc", C.c_void_p), ("last_octave", C.c_void_p), ("last_angle", C.c_void_p),
                ("has_obs", C.c_void_p), ("Tcw", C.c_float * 16), ("Tlw", C.c_float * 16),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("mb", C.c_float), ("mbf", C.c_float)]


class _ProjPoints(C.Structure):
    _fields_ = [("n_pts", C.c_int32), ("valid", C.c_void_p), ("pos", C.c_void_p), ("max_dist", C.c_void_p),
                ("min_dist", C.c_void_p), ("normal", C.c_void_p), ("desc", C.c_void_p), ("q_angle", C.c_void_p),
                ("R", C.c_float * 9), ("t", C.c_float * 3), ("Ow", C.c_float * 3), ("R2", C.c_float * 9),
                ("t2", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("bf", C.c_float), ("log_scale_factor", C.c_float), ("inv_level_sigma2", C.c_void_p), ("th", C.c_float)]


_DTYPES = dict(desc_kf=np.uint8, desc_f=np.uint8, kf_has_mp=np.uint8, angle_kf=np.float32, angle_f=np.float32,
               node_id_kf=np.int32, node_off_kf=np.int32, node_idx_kf=np.int32, node_id_f=np.int32,
               node_off_f=np.int32, node_idx_f=np.int32, kp_x=np.float32, kp_y=np.float32, kp_octave=np.int32,
               kp_angle=np.float32, u_right=np.float32, scale_factors=np.float32, grid_off=np.int32,
               grid_idx=np.int32, f_mp_state=np.uint8, track_in_view=np.uint8, pred_level=np.int32,
               view_cos=np.float32, proj_x=np.float32, proj_y=np.float32, proj_xr=np.float32, desc=np.uint8,
               has_obs=np.uint8, last_valid=np.uint8, world_pos=np.float32, last_octave=np.int32,
               last_angle=np.float32, desc1=np.uint8, desc2=np.uint8, has_mp1=np.uint8, has_mp2=np.uint8,
               angle1=np.float32, angle2=np.float32, node_id1=np.int32, node_off1=np.int32, node_idx1=np.int32,
               node_id2=np.int32, node_off2=np.int32, node_idx2=np.int32, x1=np.float32, y1=np.float32, x2=np.float32,
               y2=np.float32, u_right1=np.float32, u_right2=np.float32, octave2=np.int32, scale_factors2=np.float32,
               level_sigma2_2=np.float32, valid=np.uint8, pos=np.float32, max_dist=np.float32, min_dist=np.float32,
               normal=np.float32, q_angle=np.float32, inv_level_sigma2=np.float32)


def _fill_struct(st, d, keep):
    for name, ct in st._fields_:
        if name not in d:
            continue
        v = d[name]
        if ct is C.c_void_p:
            a = np.ascontiguousarray(v, _DTYPES[name])
            keep.append(a)
            setattr(st, name, a.ctypes.data)
        elif isinstance(v, np.ndarray) and hasattr(ct, "_length_"):
            setattr(st, name, ct(*[float(x) for x in v.reshape(-1)]))
        else:
            setattr(st, name, v.item() if hasattr(v, "item") else v)
    return st


class Matcher:
    """Mirror of ORBmatcher (include/ORBmatcher.h:37-102) over SoA snapshots."""
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30

    def __init__(self, nnratio=0.6, check_orientation=True, device=0):
        self.L = lib()
        h = C.c_void_p()
        _check(self.L.aos2_matcher_create(float(nnratio), int(bool(check_orientation)), device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.aos2_matcher_destroy(self.h)
            self.h = None

    __del__ = close

    def last_device_ms(self):
        return float(self.L.aos2_matcher_last_device_ms(self.h))

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return lib().aos2_descriptor_distance(_p(a), _p(b))

    def hamming_best2(self, q, t):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        bi = np.zeros(len(q), np.int32)
        bd = np.zeros(len(q), np.int32)
        sd = np.zeros(len(q), np.int32)
        _check(self.L.aos2_matcher_hamming_best2(self.h, _p(q), len(q), _p(t), len(t), _p(bi), _p(bd), _p(sd)))
        return bi, bd, sd

    def hamming_best2_device(self, d_q, nq, d_t, nt, d_bi, d_bd, d_sd, iters=1):
        ms = C.c_float(0)
        _check(self.L.aos2_matcher_hamming_best2_device(self.h, C.c_void_p(d_q), nq, C.c_void_p(d_t), nt,
                                                        C.c_void_p(d_bi), C.c_void_p(d_bd), C.c_void_p(d_sd),
                                                        iters, C.byref(ms)))
        return ms.value

    def SearchByBoW(self, problems):
        """problems: list of synth_bow_problem()-style dicts (or one dict). Returns [(nmatches, match_f)]."""
        single = isinstance(problems, dict)
        if single:
            problems = [problems]
        keep = []
        arr = (_BowPair * len(problems))()
        outs = []
        for i, p in enumerate(problems):
            d = dict(p)
            d["n_kf"], d["n_f"] = len(p["desc_kf"]), len(p["desc_f"])
            d["n_nodes_kf"], d["n_nodes_f"] = len(p["node_id_kf"]), len(p["node_id_f"])
            _fill_struct(arr[i], d, keep)
            outs.append(np.zeros(max(d["n_f"], 1), np.int32))
        ptrs = (C.c_void_p * len(problems))(*[o.ctypes.data for o in outs])
        nm = np.zeros(len(problems), np.int32)
        _check(self.L.aos2_matcher_search_by_bow(self.h, C.byref(arr), len(problems), ptrs, _p(nm)))
        res = [(int(nm[i]), outs[i][: len(problems[i]["desc_f"])]) for i in range(len(problems))]
        return res[0] if single else res

    def SearchByBoWDevice(self, pairs):
        """aos2_matcher_search_by_bow_device: pairs = list of dicts with DEVICE addresses (ints) desc_kf, desc_f, angle_kf,
        angle_f, sizes n_kf / n_f, and host arrays kf_has_mp, node_id_* / node_off_* / node_idx_* -> [(nmatches, match_f)]"""
        keep = []
        n = len(pairs)
        arr = (_BowPair * n)()
        outs = []
        for i, p in enumerate(pairs):
            a = arr[i]
            a.n_kf, a.n_f = int(p["n_kf"]), int(p["n_f"])
            a.desc_kf, a.desc_f, a.angle_kf, a.angle_f = int(p["desc_kf"]), int(p["desc_f"]), int(p["angle_kf"]), int(p["angle_f"])
            a.n_nodes_kf, a.n_nodes_f = len(p["node_id_kf"]), len(p["node_id_f"])
            for name in ("kf_has_mp", "node_id_kf", "node_off_kf", "node_idx_kf", "node_id_f", "node_off_f", "node_idx_f"):
                v = np.ascontiguousarray(p[name], _DTYPES[name])
                if v.size == 0:
                    v = np.zeros(1, _DTYPES[name])
                keep.append(v)
                setattr(a, name, v.ctypes.data)
            outs.append(np.zeros(max(a.n_f, 1), np.int32))
        ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        nm = np.zeros(n, np.int32)
        _check(self.L.aos2_matcher_search_by_bow_device(self.h, C.byref(arr), n, ptrs, _p(nm)))
        return [(int(nm[i]), outs[i][: int(pairs[i]["n_f"])]) for i in range(n)]

    def SearchByBoWFrames(self, n_frames, cap, kf_has_mp, match_f, nmatches, **dev):
        """aos2_matcher_search_by_bow_frames: dev = the device addresses (ints) named like aos2_bow_frames_t's fields;
        kf_has_mp uint8 [n][cap], match_f int32 [n][cap], nmatches int32 [n]: host arrays (filled in place)"""
        q = _BowFrames()
        q.n_frames, q.cap = int(n_frames), int(cap)
        for k, v in dev.items():
            setattr(q, k, int(v))
        q.kf_has_mp = kf_has_mp.ctypes.data
        _check(self.L.aos2_matcher_search_by_bow_frames(self.h, C.byref(q), _p(match_f), _p(nmatches)))

    def _kfkf(self, struct_t, fn, problems, *extra):
        single = isinstance(problems, dict)
        if single:
            problems = [problems]
        keep = []
        arr = (struct_t * len(problems))()
        outs = []
        for i, p in enumerate(problems):
            d = dict(p)
            d["n1"], d["n2"] = len(p["desc1"]), len(p["desc2"])
            d["n_nodes1"], d["n_nodes2"] = len(p["node_id1"]), len(p["node_id2"])
            if "scale_factors2" in p:
                d["n_levels2"] = len(p["scale_factors2"])
            _fill_struct(arr[i], d, keep)
            outs.append(np.zeros(max(d["n1"], 1), np.int32))
        ptrs = (C.c_void_p * len(problems))(*[o.ctypes.data for o in outs])
        nm = np.zeros(len(problems), np.int32)
        _check(fn(self.h, C.byref(arr), len(problems), *extra, ptrs, _p(nm)))
        res = [(int(nm[i]), outs[i][: len(problems[i]["desc1"])]) for i in range(len(problems))]
        return res[0] if single else res

    def SearchByBoWKF(self, problems):
        """SearchByBoW(pKF1, pKF2, vpMatches12) src/ORBmatcher.cc:522-655; synth_bow_kf_problem()-style dicts.
        Returns (nmatches, match12) per problem."""
        return self._kfkf(_BowKfPair, self.L.aos2_matcher_search_by_bow_kf, problems)

    def SearchForTriangulation(self, problems, only_stereo=False):
        """src/ORBmatcher.cc:657-823; synth_triang_problem()-style dicts. Returns (nmatches, vMatches12)."""
        return self._kfkf(_TriangPair, self.L.aos2_matcher_search_for_triangulation, problems, int(only_stereo))

    def ComputeDistinctiveDescriptors(self, off, desc):
        """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:275-340) for a CSR batch of map points."""
        off = np.ascontiguousarray(off, np.int32)
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(off) - 1
        best = np.zeros(max(n, 1), np.int32)
        _check(self.L.aos2_compute_distinctive_descriptors(self.h, n, _p(off), _p(desc), _p(best)))
        return best[:n]

    def SearchByProjection(self, f, mp, th=3.0):
        keep = []
        fv = _fill_struct(_FrameView(), f, keep)
        pm = _fill_struct(_ProjMp(), mp, keep)
        match = np.zeros(max(f["n_f"], 1), np.int32)
        n = np.zeros(1, np.int32)
        _check(self.L.aos2_matcher_search_by_projection(self.h, C.byref(fv), C.byref(pm), float(th), _p(match), _p(n)))
        return int(n[0]), match[: f["n_f"]]

    def SearchByProjectionBatch(self, frames, mps, th=3.0):
        """n independent SearchByProjection(F, vpMapPoints, th) problems in one launch -> [(nmatches, match_f)]"""
        keep = []
        n = len(frames)
        fa, pa = (_FrameView * n)(), (_ProjMp * n)()
        outs = []
        for i in range(n):
            _fill_struct(fa[i], frames[i], keep)
            _fill_struct(pa[i], mps[i], keep)
            outs.append(np.zeros(max(frames[i]["n_f"], 1), np.int32))
        ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        nm = np.zeros(n, np.int32)
        _check(self.L.aos2_matcher_search_by_projection_batch(self.h, C.byref(fa), C.byref(pa), n, float(th), ptrs, _p(nm)))
        return [(int(nm[i]), outs[i][: frames[i]["n_f"]]) for i in range(n)]

    def SearchByProjectionLast(self, cur, last, th, mono):
        keep = []
        fv = _fill_struct(_FrameView(), cur, keep)
        pl = _fill_struct(_ProjLast(), last, keep)
        match = np.zeros(max(cur["n_f"], 1), np.int32)
        n = np.zeros(1, np.int32)
        _check(self.L.aos2_matcher_search_by_projection_last(self.h, C.byref(fv), C.byref(pl), float(th), int(mono),
                                                             _p(match), _p(n)))
        return int(n[0]), match[: cur["n_f"]]


    # ---- projection family; f / p: synth_proj_gen_problem()-style dicts
    def Fuse(self, kf, p, sim3=False):
        """search part of Fuse (src/ORBmatcher.cc:825-975, sim3: :977-1100) -> (nFused, best_idx, best_dist)"""
        keep = []
        fv = _fill_struct(_FrameView(), kf, keep)
        pp = _fill_struct(_ProjPoints(), p, keep)
        bi, bd = np.zeros(max(p["n_pts"], 1), np.int32), np.zeros(max(p["n_pts"], 1), np.int32)
        n = np.zeros(1, np.int32)
        _check(self.L.aos2_matcher_fuse(self.h, C.byref(fv), C.byref(pp), int(sim3), _p(bi), _p(bd), _p(n)))
        return int(n[0]), bi[: p["n_pts"]], bd[: p["n_pts"]]

    def SearchByProjectionKF(self, kf, p):
        """SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) :290-403 -> (nmatches, match_f)"""
        keep = []
        fv = _fill_struct(_FrameView(), kf, keep)
        pp = _fill_struct(_ProjPoints(), p, keep)
        match, n = np.zeros(max(kf["n_f"], 1), np.int32), np.zeros(1, np.int32)
        _check(self.L.aos2_matcher_search_by_projection_kf(self.h, C.byref(fv), C.byref(pp), _p(match), _p(n)))
        return int(n[0]), match[: kf["n_f"]]

    def SearchBySim3(self, kf1, kf2, p12, p21):
        """SearchBySim3 :1102-1326 -> (nFound, match12)"""
        keep = []
        a, b = _fill_struct(_FrameView(), kf1, keep), _fill_struct(_FrameView(), kf2, keep)
        c, d = _fill_struct(_ProjPoints(), p12, keep), _fill_struct(_ProjPoints(), p21, keep)
        match, n = np.zeros(max(p12["n_pts"], 1), np.int32), np.zeros(1, np.int32)
        _check(self.L.aos2_matcher_search_by_sim3(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), _p(match), _p(n)))
        return int(n[0]), match[: p12["n_pts"]]

    def AssignFeaturesToGrid(self, kp_x, kp_y, min_x, min_y, grid_w_inv, grid_h_inv):
        """Frame::AssignFeaturesToGrid (src/Frame.cc:259-274) -> (grid_off[3073], grid_idx)"""
        kp_x, kp_y = np.ascontiguousarray(kp_x, np.float32), np.ascontiguousarray(kp_y, np.float32)
        off = np.zeros(64 * 48 + 1, np.int32)
        idx = np.zeros(max(len(kp_x), 1), np.int32)
        n = C.c_int32(0)
        _check(self.L.aos2_frame_assign_features_to_grid(self.h, len(kp_x), _p(kp_x), _p(kp_y), float(min_x), float(min_y),
                                                         float(grid_w_inv), float(grid_h_inv), _p(off), _p(idx), C.byref(n)))
        return off, idx[: n.value]

    def ComputeStereoFromRGBD(self, kp_x, kp_y, kpun_x, depth_img, mbf):
        """Frame::ComputeStereoFromRGBD (src/Frame.cc:672-693) -> (mvuRight, mvDepth)"""
        kp_x, kp_y, kpun_x = (np.ascontiguousarray(a, np.float32) for a in (kp_x, kp_y, kpun_x))
        depth_img = np.ascontiguousarray(depth_img, np.float32)
        ur, dp = np.zeros(max(len(kp_x), 1), np.float32), np.zeros(max(len(kp_x), 1), np.float32)
        _check(self.L.aos2_frame_stereo_from_rgbd(self.h, len(kp_x), _p(kp_x), _p(kp_y), _p(kpun_x), _p(depth_img),
                                                  depth_img.shape[1], depth_img.shape[0], depth_img.shape[1], float(mbf),
                                                  _p(ur), _p(dp)))
        return ur[: len(kp_x)], dp[: len(kp_x)]

    def isInFrustum(self, frame, p, viewing_cos_limit=0.5):
        """Frame::isInFrustum (src/Frame.cc:298-354) for all points of p -> dict with the aos2_proj_mp_t arrays"""
        keep = []
        pp = _fill_struct(_ProjPoints(), p, keep)
        n = p["n_pts"]
        iv = np.zeros(max(n, 1), np.uint8)
        px, py, pr, vc = (np.zeros(max(n, 1), np.float32) for _ in range(4))
        lv = np.zeros(max(n, 1), np.int32)
        _check(self.L.aos2_frame_is_in_frustum(self.h, C.byref(pp), float(frame["min_x"]), float(frame["max_x"]),
                                               float(frame["min_y"]), float(frame["max_y"]), int(frame["n_levels"]),
                                               float(viewing_cos_limit), _p(iv), _p(px), _p(py), _p(pr), _p(lv), _p(vc)))
        return dict(track_in_view=iv[:n], proj_x=px[:n], proj_y=py[:n], proj_xr=pr[:n], pred_level=lv[:n], view_cos=vc[:n])

    def SearchForInitialization(self, f2, q, window_size=100):
        """SearchForInitialization :405-520; q = dict(desc1, octave1, angle1, prev_xy) -> (nmatches, vnMatches12)"""
        keep = []
        fv = _fill_struct(_FrameView(), f2, keep)
        d1 = np.ascontiguousarray(q["desc1"], np.uint8)
        o1 = np.ascontiguousarray(q["octave1"], np.int32)
        a1 = np.ascontiguousarray(q["angle1"], np.float32)
        pv = np.ascontiguousarray(q["prev_xy"], np.float32)
        match, n = np.zeros(max(len(d1), 1), np.int32), np.zeros(1, np.int32)
        _check(self.L.aos2_matcher_search_for_initialization(self.h, C.byref(fv), len(d1), _p(d1), _p(o1), _p(a1), _p(pv),
                                                             int(window_size), _p(match), _p(n)))
        return int(n[0]), match[: len(d1)]

    def SearchByProjectionReloc(self, frame, p, orb_dist=100):
        """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) :1472-1599 -> (nmatches, match_f)"""
        keep = []
        fv = _fill_struct(_FrameView(), frame, keep)
        pp = _fill_struct(_ProjPoints(), p, keep)
        match, n = np.zeros(max(frame["n_f"], 1), np.int32), np.zeros(1, np.int32)
        _check(self.L.aos2_matcher_search_by_projection_reloc(self.h, C.byref(fv), C.byref(pp), int(orb_dist), _p(match), _p(n)))
        return int(n[0]), match[: frame["n_f"]]


# ------------------------------------------------------------------------------------------ local BA
class _LbaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32),
                ("pose_Tcw", C.c_void_p), ("pose_fixed", C.c_void_p), ("pose_id", C.c_void_p),
                ("point_xyz", C.c_void_p), ("point_id", C.c_void_p), ("edge_pose", C.c_void_p),
                ("edge_point", C.c_void_p), ("edge_obs", C.c_void_p), ("edge_stereo", C.c_void_p),
                ("edge_inv_sigma2", C.c_void_p),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float),
                ("stop_flag", C.c_void_p), ("iters_first", C.c_int32), ("iters_second", C.c_int32)]


class _LbaResult(C.Structure):
    _fields_ = [("pose_Tcw", C.c_void_p), ("point_xyz", C.c_void_p), ("edge_outlier", C.c_void_p),
                ("edge_chi2", C.c_void_p), ("iters_done_first", C.c_int32), ("iters_done_second", C.c_int32),
                ("final_chi2", C.c_double), ("final_lambda", C.c_double), ("ms_device", C.c_float),
                ("status", C.c_int32), ("trials_first", C.c_int32), ("trials_second", C.c_int32),
                ("polls", C.c_int32), ("stop_poll", C.c_int32)]


class _PoseProblem(C.Structure):
    _fields_ = [("n", C.c_int32), ("Xw", C.c_void_p), ("obs", C.c_void_p), ("stereo", C.c_void_p),
                ("inv_sigma2", C.c_void_p), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("bf", C.c_float), ("Tcw", C.c_float * 16)]


class _PoseResult(C.Structure):
    _fields_ = [("Tcw", C.c_float * 16), ("outlier", C.c_void_p), ("n_bad", C.c_int32), ("n_inliers", C.c_int32)]


def lba_host_phase(probs, threads):
    """aos2_lba_debug_host_phase: (build ms, stage ms) of the host part of a LocalBA batch; needs no device"""
    n = len(probs)
    keep = []
    S = (_LbaProblem * n)()
    for i, prob in enumerate(probs):
        s = S[i]
        s.n_poses, s.n_points, s.n_edges = prob["n_poses"], prob["n_points"], prob["n_edges"]
        for name, dt in (("pose_Tcw", np.float32), ("pose_fixed", np.uint8), ("pose_id", np.int64), ("point_xyz", np.float32),
                         ("point_id", np.int64), ("edge_pose", np.int32), ("edge_point", np.int32), ("edge_obs", np.float32),
                         ("edge_stereo", np.uint8), ("edge_inv_sigma2", np.float32)):
            a = np.ascontiguousarray(prob[name], dt)
            keep.append(a)
            setattr(s, name, a.ctypes.data)
    b, st = C.c_double(), C.c_double()
    _check(lib().aos2_lba_debug_host_phase(C.byref(S), n, int(threads), C.byref(b), C.byref(st)))
    return float(b.value), float(st.value)


class LocalBA:
    """Optimizer::LocalBundleAdjustment numerical part (include/Optimizer.h:45, src/Optimizer.cc:454-779)."""

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        _check(self.L.aos2_lba_create(device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.aos2_lba_destroy(self.h)
            self.h = None

    __del__ = close

    def _fill(self, s, r, prob, stop_flag, iters, keep):
        s.n_poses, s.n_points, s.n_edges = prob["n_poses"], prob["n_points"], prob["n_edges"]
        for name, dt in (("pose_Tcw", np.float32), ("pose_fixed", np.uint8), ("pose_id", np.int64),
                         ("point_xyz", np.float32), ("point_id", np.int64), ("edge_pose", np.int32),
                         ("edge_point", np.int32), ("edge_obs", np.float32), ("edge_stereo", np.uint8),
                         ("edge_inv_sigma2", np.float32)):
            a = np.ascontiguousarray(prob[name], dt)
            keep.append(a)
            setattr(s, name, a.ctypes.data)
        s.fx, s.fy, s.cx, s.cy, s.bf = (float(np.float32(prob[k])) for k in ("fx", "fy", "cx", "cy", "bf"))
        if stop_flag is not None:
            keep.append(stop_flag)
            s.stop_flag = stop_flag.ctypes.data
        s.iters_first, s.iters_second = iters
        Tout = np.zeros((s.n_poses, 16), np.float32)
        Pout = np.zeros((s.n_points, 3), np.float32)
        outl = np.zeros(s.n_edges, np.uint8)
        chi2 = np.zeros(s.n_edges, np.float64)
        r.pose_Tcw, r.point_xyz, r.edge_outlier, r.edge_chi2 = Tout.ctypes.data, Pout.ctypes.data, outl.ctypes.data, chi2.ctypes.data
        return Tout, Pout, outl, chi2

    @staticmethod
    def _result(r, arrs):
        Tout, Pout, outl, chi2 = arrs
        return dict(status=r.status, pose_Tcw=Tout, point_xyz=Pout, edge_outlier=outl, edge_chi2=chi2,
                    iters=(r.iters_done_first, r.iters_done_second), trials=(r.trials_first, r.trials_second),
                    final_chi2=r.final_chi2, final_lambda=r.final_lambda, ms_device=r.ms_device,
                    polls=r.polls, stop_poll=r.stop_poll)

    def LocalBundleAdjustment(self, prob, stop_flag=None, iters=(5, 10)):
        keep = []
        s, r = _LbaProblem(), _LbaResult()
        arrs = self._fill(s, r, prob, stop_flag, iters, keep)
        _check(self.L.aos2_lba_solve(self.h, C.byref(s), C.byref(r)), ok=(AOS2_OK, AOS2_ERR_STOPPED))
        return self._result(r, arrs)

    def LocalBundleAdjustmentBatch(self, probs, stop_flags=None, iters=(5, 10)):
        """aos2_lba_solve_batch: independent windows in one call."""
        n = len(probs)
        keep = []
        S, R = (_LbaProblem * n)(), (_LbaResult * n)()
        arrs = [self._fill(S[i], R[i], probs[i], None if stop_flags is None else stop_flags[i], iters, keep) for i in range(n)]
        _check(self.L.aos2_lba_solve_batch(self.h, C.byref(S), C.byref(R), n))
        return [self._result(R[i], arrs[i]) for i in range(n)]

    def prepare_batch(self, probs, iters=(5, 10), want_chi2=False):
        """ctypes problem / result arrays for repeated aos2_lba_solve_batch calls on the same inputs (bench harness)"""
        n = len(probs)
        keep = []
        S, R = (_LbaProblem * n)(), (_LbaResult * n)()
        arrs = [self._fill(S[i], R[i], probs[i], None, iters, keep) for i in range(n)]
        if not want_chi2:
            for i in range(n):
                R[i].edge_chi2 = None
        return dict(S=S, R=R, arrs=arrs, keep=keep, n=n)

    def solve_prepared(self, prep):
        _check(self.L.aos2_lba_solve_batch(self.h, C.byref(prep["S"]), C.byref(prep["R"]), prep["n"]))
        return prep["R"]

    def set_host_threads(self, n):
        _check(self.L.aos2_lba_set_host_threads(self.h, int(n)))

    def set_window_groups(self, n):
        """0 = default, 1 = one program for all windows of a batch, 2 = two staggered window groups (include/aos2.h)"""
        _check(self.L.aos2_lba_set_window_groups(self.h, int(n)))

    def last_program(self):
        """(trial slots enqueued for every window, host rounds) of the last solve"""
        a, b = C.c_int32(), C.c_int32()
        _check(self.L.aos2_lba_last_program(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def last_window_slots(self):
        """trial slots of the last solve summed over the windows each round covered (include/aos2.h)"""
        a = C.c_int64()
        _check(self.L.aos2_lba_last_window_slots(self.h, C.byref(a)))
        return int(a.value)

    def debug_stop_at_poll(self, poll):
        _check(self.L.aos2_lba_debug_stop_at_poll(self.h, int(poll)))

    def PoseOptimization(self, problems):
        """int Optimizer::PoseOptimization(Frame*) (src/Optimizer.cc:239-452) for one dict or a list of
        synth_pose_problem()-style dicts (one workgroup per frame, one launch)."""
        single = isinstance(problems, dict)
        if single:
            problems = [problems]
        keep = []
        P = (_PoseProblem * len(problems))()
        R = (_PoseResult * len(problems))()
        outs = []
        for i, p in enumerate(problems):
            P[i].n = int(p["n"])
            for name, dt in (("Xw", np.float32), ("obs", np.float32), ("stereo", np.uint8), ("inv_sigma2", np.float32)):
                a = np.ascontiguousarray(p[name], dt)
                keep.append(a)
                setattr(P[i], name, a.ctypes.data)
            P[i].fx, P[i].fy, P[i].cx, P[i].cy, P[i].bf = (float(np.float32(p[k])) for k in ("fx", "fy", "cx", "cy", "bf"))
            P[i].Tcw = (C.c_float * 16)(*[float(x) for x in np.asarray(p["Tcw"], np.float32).reshape(-1)])
            o = np.zeros(max(P[i].n, 1), np.uint8)
            outs.append(o)
            R[i].outlier = o.ctypes.data
        _check(self.L.aos2_pose_optimization(self.h, C.byref(P), C.byref(R), len(problems)))
        res = [dict(n_inliers=R[i].n_inliers, n_bad=R[i].n_bad, outlier=outs[i][: P[i].n].copy(),
                    Tcw=np.array(list(R[i].Tcw), np.float32)) for i in range(len(problems))]
        return res[0] if single else res

    def pose_last_device_ms(self):
        return float(self.L.aos2_pose_optimization_last_device_ms(self.h))


# ---------------------------------------------------------------------------------------------------------------
# Device-resident frame batches (include/aos2.h: aos2_frames_*).  Device arrays are passed as raw pointers (torch
# tensors' data_ptr()); this wrapper owns nothing but the handle.
class _MapPointsDev(C.Structure):
    _fields_ = [("n", C.c_int32), ("pos", C.c_void_p), ("desc", C.c_void_p), ("has_obs", C.c_void_p),
                ("normal", C.c_void_p), ("min_dist", C.c_void_p), ("max_dist", C.c_void_p)]


def map_points_dev(n, pos, desc, has_obs, normal=0, min_dist=0, max_dist=0):
    """aos2_map_points_dev_t from device pointers (ints)"""
    t = _MapPointsDev()
    t.n, t.pos, t.desc, t.has_obs, t.normal, t.min_dist, t.max_dist = int(n), pos, desc, has_obs, normal or None, min_dist or None, max_dist or None
    return t


def frame_image_bounds(w, h, fx, fy, cx, cy, dist):
    """Frame::ComputeImageBounds (src/Frame.cc:463-493) -> (mnMinX, mnMaxX, mnMinY, mnMaxY)"""
    d = np.zeros(5, np.float32)
    d[: len(dist)] = np.asarray(dist, np.float32)
    out = np.zeros(4, np.float32)
    _check(lib().aos2_frame_image_bounds(int(w), int(h), float(fx), float(fy), float(cx), float(cy), _p(d), _p(out)))
    return out


class _FramesTriang(C.Structure):
    _fields_ = [("n_pairs", C.c_int32)] + [(k, C.c_void_p) for k in ("kf1", "kf2", "F12", "epipole", "d_node_of1", "d_fv_node1", "d_fv_off1",
                                                                      "d_fv_idx1", "d_n_fv1", "d_fv_node2", "d_fv_off2", "d_fv_idx2", "d_n_fv2")]


class Frames:
    MAP_POINTS, OUTLIER, TCW, U_RIGHT, DEPTH, GRID_OFF, GRID_IDX, KEYS_UN_X, KEYS_UN_Y = range(9)

    def __init__(self, batch, cap, device=0):
        self.L = lib()
        h = C.c_void_p()
        _check(self.L.aos2_frames_create(device, batch, cap, C.byref(h)))
        self.h, self.batch, self.cap = h, batch, cap

    def close(self):
        if getattr(self, "h", None):
            self.L.aos2_frames_destroy(self.h)
            self.h = None

    __del__ = close

    def build(self, extractor, d_kps, d_desc, d_n, w, h, d_depth, fx, fy, cx, cy, mbf, depth_stride=None, depth_image_stride=None):
        _check(self.L.aos2_frames_build(self.h, extractor.h, self.batch, d_kps, d_desc, d_n, self.cap, w, h, d_depth or None,
                                        depth_stride or w, depth_image_stride or w * h, fx, fy, cx, cy, mbf))

    def build_stereo(self, extractor_left, d_kps, d_desc, d_n, w, h, d_u_right, d_depth_kp, fx, fy, cx, cy, mbf):
        """Frame::Frame(imLeft, imRight, ...) after ExtractORB x 2 + ComputeStereoMatches (src/Frame.cc:57-113)"""
        _check(self.L.aos2_frames_build_stereo(self.h, extractor_left.h, self.batch, d_kps, d_desc, d_n, self.cap, w, h, d_u_right, d_depth_kp,
                                               fx, fy, cx, cy, mbf))

    def set_pose(self, d_Tcw):
        _check(self.L.aos2_frames_set_pose(self.h, d_Tcw))

    def set_distortion(self, dist):
        """mDistCoef = k1 k2 p1 p2 k3 for the following build() calls"""
        d = list(dist) + [0.0] * (5 - len(dist))
        _check(self.L.aos2_frames_set_distortion(self.h, *[float(v) for v in d[:5]]))

    def set_map_points(self, mp, table, outlier=None):
        mp = np.ascontiguousarray(mp, np.int32).reshape(self.batch, self.cap)
        if outlier is not None:
            outlier = np.ascontiguousarray(outlier, np.uint8).reshape(self.batch, self.cap)
        _check(self.L.aos2_frames_set_map_points(self.h, _p(mp), None if outlier is None else _p(outlier), C.byref(table)))

    def get(self, what):
        B, cap = self.batch, self.cap
        shape, dt = {0: ((B, cap), np.int32), 1: ((B, cap), np.uint8), 2: ((B, 16), np.float32), 3: ((B, cap), np.float32),
                     4: ((B, cap), np.float32), 5: ((B, 64 * 48 + 1), np.int32), 6: ((B, cap), np.int32), 7: ((B, cap), np.float32),
                     8: ((B, cap), np.float32)}[what]
        a = np.zeros(shape, dt)
        _check(self.L.aos2_frames_get(self.h, what, _p(a), a.nbytes))
        return a

    def SearchByProjectionLast(self, last, table, th, mono=False, check_orientation=True, d_nmatches=0):
        _check(self.L.aos2_frames_search_by_projection_last(self.h, last.h, C.byref(table), th, int(mono), int(check_orientation),
                                                            d_nmatches or None))

    def PoseOptimization(self, table, d_inliers=0):
        _check(self.L.aos2_frames_pose_optimization(self.h, C.byref(table), d_inliers or None))

    def discard_outliers(self):
        _check(self.L.aos2_frames_discard_outliers(self.h))

    def SearchLocalPoints(self, table, d_local, n_local, th, nnratio, d_nmatches=0):
        _check(self.L.aos2_frames_search_local_points(self.h, C.byref(table), d_local, n_local, th, nnratio, d_nmatches or None))

    def set_async_keyframe_calls(self, on=True):
        """SearchForTriangulation / Fuse of this handle return after enqueueing (include/aos2.h); wait() completes them"""
        self.L.aos2_frames_set_async_keyframe_calls.argtypes = [C.c_void_p, C.c_int]
        _check(self.L.aos2_frames_set_async_keyframe_calls(self.h, 1 if on else 0))

    def wait(self):
        _check(self.L.aos2_frames_wait(self.h))

    def stream(self):
        return self.L.aos2_frames_stream(self.h)

    KEYS_ANGLE, KEYS_OCTAVE = 9, 10

    def device_ptr(self, what):
        """device address (int) of a member array [batch][cap]"""
        return self.L.aos2_frames_device_ptr(self.h, int(what)) or 0

    def SearchForTriangulation(self, other, kf1, kf2, F12, epipole, d_node_of1, fv1, fv2, d_match12, d_nmatches,
                               only_stereo=False, check_orientation=True):
        """aos2_frames_search_for_triangulation: pairs (frame kf1[p] of this batch, frame kf2[p] of `other`); kf1 / kf2 int32 [n], F12 float32
        [n][9], epipole float32 [n][2]: host arrays; d_node_of1 and fv1 / fv2 = (fv_node, fv_off, fv_idx, n_fv): device addresses (ints);
        results in d_match12 [n][cap], d_nmatches [n]"""
        kf1, kf2 = np.ascontiguousarray(kf1, np.int32), np.ascontiguousarray(kf2, np.int32)
        F12, epipole = np.ascontiguousarray(F12, np.float32), np.ascontiguousarray(epipole, np.float32)
        q = _FramesTriang()
        q.n_pairs = len(kf1)
        q.kf1, q.kf2, q.F12, q.epipole = kf1.ctypes.data, kf2.ctypes.data, F12.ctypes.data, epipole.ctypes.data
        q.d_node_of1 = int(d_node_of1)
        q.d_fv_node1, q.d_fv_off1, q.d_fv_idx1, q.d_n_fv1 = (int(x) for x in fv1)
        q.d_fv_node2, q.d_fv_off2, q.d_fv_idx2, q.d_n_fv2 = (int(x) for x in fv2)
        _check(self.L.aos2_frames_search_for_triangulation(self.h, other.h, C.byref(q), int(only_stereo), int(check_orientation), d_match12, d_nmatches))

    def Fuse(self, table, target, d_rows, n_pts, th, d_best_idx, d_best_dist):
        """aos2_frames_fuse: target int32 [n] host (frames of this batch), d_rows device [n][n_pts] table rows (-1 = rejected by the loop head)"""
        target = np.ascontiguousarray(target, np.int32)
        _check(self.L.aos2_frames_fuse(self.h, C.byref(table), len(target), int(n_pts), target.ctypes.data, d_rows, float(th), d_best_idx, d_best_dist))

    def wait_for_stream(self, hip_stream=None):
        """device-side ordering: what is enqueued on the batch from now on runs behind the work on `hip_stream` so far"""
        _check(self.L.aos2_frames_wait_for_stream(self.h, hip_stream))

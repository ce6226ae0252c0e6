"""Harness (tests/, bench.py): the device-resident per-frame tracking chain over a scenario.tracking_scenario().

    ORBextractor::operator() -> Frame::Frame -> SearchByProjection(Current, Last) -> PoseOptimization -> outlier discard
    -> SearchLocalPoints -> PoseOptimization                       (Tracking.cc:862-870, 963-1027, 1302-1352, 1039)

PyTorch is only the owner of the device buffers; every stage is one C-ABI call of include/aos2.h.  `step()` enqueues
the whole chain for the batch and returns; `wait()` completes it.
"""
from __future__ import annotations

import numpy as np

from . import capi, scenario


HIP_D2D = 3   # hipMemcpyDeviceToDevice


class TrackingChain:
    def __init__(self, scen: dict, device: int = 0, n_local: int = 1500, th_last: float = 15.0, th_local: float = 3.0,
                 nnratio_local: float = 0.8):
        import torch
        self.torch = torch
        self.scen, self.device = scen, device
        self.dev = torch.device("cuda", device)
        self.th_last, self.th_local, self.nnratio_local, self.n_local = th_last, th_local, nnratio_local, n_local
        B, W, H = scen["batch"], scen["w"], scen["h"]
        self.B, self.W, self.H = B, W, H
        self.ex = capi.Extractor(nfeatures=scen["nfeatures"], device=device)
        self.ex_setup = capi.Extractor(nfeatures=scen["nfeatures"], device=device)
        self.cap = cap = self.ex.max_keypoints_for(W, H)
        idx = scen["index"]
        t = torch
        self.d_cur = t.from_numpy(scen["cur"][idx]).to(self.dev)
        self.d_depth = t.from_numpy(scen["depth_cur"][idx]).to(self.dev)
        self.d_guess = t.from_numpy(np.ascontiguousarray(scen["Tcw_guess"][idx].reshape(B, 16))).to(self.dev)
        self.d_kps = t.zeros((B, cap, 7), dtype=t.float32, device=self.dev)
        self.d_desc = t.zeros((B, cap, 32), dtype=t.uint8, device=self.dev)
        self.d_n = t.zeros((B,), dtype=t.int32, device=self.dev)
        self.d_nm = t.zeros((4, B), dtype=t.int32, device=self.dev)   # nmatches last / inliers 1 / nmatches local / inliers 2
        self.cur = capi.Frames(B, cap, device)
        if scen.get("dist") is not None:
            self.cur.set_distortion(scen["dist"])
        self._setup_last()

    # ---- LastFrame batch, MapPoint table and local lists (setup, untimed): built from the extractor's keypoints
    def _setup_last(self):
        t, scen, B, cap = self.torch, self.scen, self.B, self.cap
        nu, idx = scen["n_unique"], scen["index"]
        lk = self.ex_setup.extract_batch(scen["last"])
        ok = self.ex_setup.extract_batch(scen["old"])
        self.host_last = lk
        m = scenario.build_map(scen, [k for k, _ in lk], [d for _, d in lk], [k for k, _ in ok], [d for _, d in ok],
                               self.ex.GetScaleFactors(), n_local=self.n_local)
        self.map = m
        tb = m["table"]
        self.d_table = {k: t.from_numpy(np.ascontiguousarray(tb[k])).to(self.dev) for k in ("pos", "desc", "has_obs", "normal", "min_dist", "max_dist")}
        self.table = capi.map_points_dev(tb["n"], *(self.d_table[k].data_ptr() for k in ("pos", "desc", "has_obs", "normal", "min_dist", "max_dist")))
        # the LastFrame batch holds the last views' extraction (tiled like the current frames)
        self.d_last_img = t.from_numpy(scen["last"][idx]).to(self.dev)
        self.d_last_depth = t.from_numpy(scen["depth_last"][idx]).to(self.dev)
        self.dl_kps = t.zeros((B, cap, 7), dtype=t.float32, device=self.dev)
        self.dl_desc = t.zeros((B, cap, 32), dtype=t.uint8, device=self.dev)
        self.dl_n = t.zeros((B,), dtype=t.int32, device=self.dev)
        t.cuda.synchronize()   # (torch's fills run on torch's stream, the library on its own: the fills first)
        W, H = self.W, self.H
        self.ex_setup.extract_batch_device(self.d_last_img.data_ptr(), B, W, H, W, W * H, self.dl_kps.data_ptr(), self.dl_desc.data_ptr(),
                                           cap, self.dl_n.data_ptr())
        self.last = capi.Frames(B, cap, self.device)
        if scen.get("dist") is not None:
            self.last.set_distortion(scen["dist"])
        s = scen
        self.last.build(self.ex_setup, self.dl_kps.data_ptr(), self.dl_desc.data_ptr(), self.dl_n.data_ptr(), W, H,
                        self.d_last_depth.data_ptr(), float(s["fx"]), float(s["fy"]), float(s["cx"]), float(s["cy"]), float(s["mbf"]))
        self.d_Tlw = t.from_numpy(np.ascontiguousarray(scen["Tlw"][idx].reshape(B, 16))).to(self.dev)
        self.last.set_pose(self.d_Tlw.data_ptr())
        mp = np.full((B, cap), -1, np.int32)
        w_ = m["mp_last"].shape[1]
        mp[:, :w_] = m["mp_last"][idx]
        self.last_outlier = np.zeros((B, cap), np.uint8)
        self.last_outlier[:, :w_] = m["outlier_last"][idx]
        self.last.set_map_points(mp, self.table, self.last_outlier)
        self.last_mp = mp
        self.d_local = t.from_numpy(np.ascontiguousarray(m["local"][idx])).to(self.dev)
        self.last.wait()

    def enqueue_extract(self):
        """ExtractORB (src/Frame.cc:276-282) for the batch, asynchronous"""
        B, W, H, cap = self.B, self.W, self.H, self.cap
        self.ex.extract_batch_device_async(self.d_cur.data_ptr(), B, W, H, W, W * H, self.d_kps.data_ptr(), self.d_desc.data_ptr(), cap,
                                           self.d_n.data_ptr())

    def enqueue_build(self):
        """the rest of Frame::Frame(imGray, imDepth, ...) (src/Frame.cc:116-170), ordered behind the extraction on the device"""
        W, H, s = self.W, self.H, self.scen
        self.cur.build(self.ex, self.d_kps.data_ptr(), self.d_desc.data_ptr(), self.d_n.data_ptr(), W, H, self.d_depth.data_ptr(),
                       float(s["fx"]), float(s["fy"]), float(s["cx"]), float(s["cy"]), float(s["mbf"]))

    def wait_extract(self):
        self.ex.wait()

    def step(self):
        """enqueue the chain for the batch"""
        c = self.cur
        c.set_pose(self.d_guess.data_ptr())   # mVelocity * mLastFrame.mTcw is known before the image: its copy runs beside the extraction
        if self.hb is not None:
            self._images_from_host()
        self.enqueue_extract()
        self.enqueue_build()
        c.SearchByProjectionLast(self.last, self.table, self.th_last, mono=False, check_orientation=True, d_nmatches=self.d_nm[0].data_ptr())
        c.PoseOptimization(self.table, self.d_nm[1].data_ptr())
        c.discard_outliers()
        c.SearchLocalPoints(self.table, self.d_local.data_ptr(), self.n_local, self.th_local, self.nnratio_local, self.d_nm[2].data_ptr())
        c.PoseOptimization(self.table, self.d_nm[3].data_ptr())
        if self.hb is not None:
            self._results_to_host()

    # ---- the reference's host boundary (bench.py --host-images): the images of a step come from page-locked HOST memory (the
    # reference's operator() takes a host cv::Mat, src/Frame.cc:276-282) and what Tracking reads of the Frame afterwards -- mvKeys,
    # mDescriptors, N, mvuRight, mvDepth, mvpMapPoints, mvbOutlier, mTcw, the match counts -- lands in page-locked host arrays:
    # one copy up on a stream of the harness (the extraction waits for it on the device), the copies down on the Frame batch's
    # stream behind the second PoseOptimization (wait() covers them).
    hb = None

    def enable_host_boundary(self, on=True, prefetch=False):
        """prefetch: the images of a step were copied up during the step BEFORE (into a staging buffer on the device, the way a frame
        grabber double-buffers); the step starts with a device-to-device copy staging -> image buffer and then starts the next
        upload, so that the 157 MB of a 512-frame step travel beside the previous step's kernels instead of in front of this step's"""
        t, F = self.torch, capi.Frames
        if not on:
            self.hb = None
            return
        H = capi.hip_runtime()
        s_io = capi.C.c_void_p()
        capi._check(H.hipStreamCreateWithFlags(capi.C.byref(s_io), 1))   # hipStreamNonBlocking
        B, cap = self.B, self.cap
        imgs = [("cur", self.d_cur, t.from_numpy(self.scen["cur"][self.scen["index"]]).pin_memory())]
        if hasattr(self, "d_right"):
            imgs.append(("right", self.d_right, t.from_numpy(self.scen["right_cur"][self.scen["index"]]).pin_memory()))
        pin = lambda shape, dt: t.zeros(shape, dtype=dt).pin_memory()   # noqa: E731
        c = self.cur
        down = [("kps", self.d_kps.data_ptr(), pin((B, cap, 7), t.float32)), ("desc", self.d_desc.data_ptr(), pin((B, cap, 32), t.uint8)),
                ("n", self.d_n.data_ptr(), pin((B,), t.int32)), ("nm", self.d_nm.data_ptr(), pin((4, B), t.int32))]
        self.hb = dict(s_io=s_io, imgs=imgs, down=down, members=None,
                       pins=dict(mp=pin((B, cap), t.int32), outlier=pin((B, cap), t.uint8), Tcw=pin((B, 16), t.float32),
                                 u_right=pin((B, cap), t.float32), depth=pin((B, cap), t.float32)))
        self.hb["up_bytes"] = sum(h.numel() * h.element_size() for _, _, h in imgs)
        self.hb["stage"] = None
        if prefetch:
            self.hb["stage"] = [t.empty_like(d) for _, d, _ in imgs]
            t.cuda.synchronize()
            for st_, (_, _, h) in zip(self.hb["stage"], imgs):   # the first step's images
                capi._check(H.hipMemcpyAsync(st_.data_ptr(), h.data_ptr(), h.numel() * h.element_size(), capi.HIP_H2D, s_io))

    def _images_from_host(self):
        H, hb = capi.hip_runtime(), self.hb
        if hb["stage"] is None:
            for name, d, h in hb["imgs"]:
                capi._check(H.hipMemcpyAsync(d.data_ptr(), h.data_ptr(), h.numel() * h.element_size(), capi.HIP_H2D, hb["s_io"]))
        else:   # (all on the one stream: staging -> images, [the extraction waits for this point], next images -> staging)
            for st_, (name, d, h) in zip(hb["stage"], hb["imgs"]):
                capi._check(H.hipMemcpyAsync(d.data_ptr(), st_.data_ptr(), h.numel() * h.element_size(), HIP_D2D, hb["s_io"]))
        self.ex.wait_for_stream(hb["s_io"])
        if hasattr(self, "ex_r"):
            self.ex_r.wait_for_stream(hb["s_io"])
        if hb["stage"] is not None:
            for st_, (name, d, h) in zip(hb["stage"], hb["imgs"]):
                capi._check(H.hipMemcpyAsync(st_.data_ptr(), h.data_ptr(), h.numel() * h.element_size(), capi.HIP_H2D, hb["s_io"]))

    def _results_to_host(self):
        H, hb, F, c = capi.hip_runtime(), self.hb, capi.Frames, self.cur
        if hb["members"] is None:   # (the member arrays exist once the batch was built: their device addresses do not change afterwards)
            ids = dict(mp=F.MAP_POINTS, outlier=F.OUTLIER, Tcw=F.TCW, u_right=F.U_RIGHT, depth=F.DEPTH)
            hb["members"] = [(k, c.device_ptr(v), hb["pins"][k]) for k, v in ids.items()]
            if not all(p for _, p, _ in hb["members"]):
                raise RuntimeError("host boundary: run one step() before enabling it (the Frame batch has no members yet)")
            hb["down_bytes"] = sum(h.numel() * h.element_size() for _, _, h in hb["down"] + hb["members"])
        q = c.stream()
        for _, d, h in hb["down"] + hb["members"]:
            capi._check(H.hipMemcpyAsync(h.data_ptr(), d, h.numel() * h.element_size(), capi.HIP_D2H, q))

    def wait(self):
        self.cur.wait()
        self.ex.wait()

    # ---- the step as ONE launch (include/aos2.h "Replay of a fixed call sequence"): the calls of step() recorded once, replayed per
    # frame.  The buffers the recording names (image, pose guess, local-map rows, outputs) are the chain's own: new contents, same places.
    def capture_step(self):
        if self.hb is not None:
            raise RuntimeError("capture_step: the host-boundary copies are not part of the recorded step")
        self.step()
        self.wait()   # (every handle has allocated for this shape)
        q = self.cur.stream()
        with capi.Graph.capture(q) as gr:
            for ex in (self.ex, getattr(self, "ex_r", None)):   # the extractors' streams join the recording; Frame::build's device-side
                if ex is not None:                               # wait on the extraction ends their part
                    ex.wait_for_stream(q)
            self.step()
        self.graph = gr
        return gr

    def step_graph(self):
        self.graph.launch()

    # ---- ONE sequence, frame after frame, with the NEXT image's ExtractORB already running: the extraction of frame t + 1 depends on
    # nothing of frame t (Tracking::GrabImageRGBD builds the Frame from the image alone, src/Tracking.cc:207-235), so it is enqueued
    # on the extractor's stream as soon as frame t's Frame::Frame has taken its own extraction, beside frame t's searches and
    # PoseOptimizations.  Two output buffer sets; the frame's latency is unchanged, the period is max(extraction, tracking).
    def step_pipelined(self):
        t = self.torch
        if not hasattr(self, "_sets"):
            self._sets = [(self.d_kps, self.d_desc, self.d_n), (t.zeros_like(self.d_kps), t.zeros_like(self.d_desc), t.zeros_like(self.d_n))]
            t.cuda.synchronize()
            self._k = 0
            self._extract_into(self._sets[0])
        B, W, H, s, c = self.B, self.W, self.H, self.scen, self.cur
        kps, desc, n = self._sets[self._k]
        c.set_pose(self.d_guess.data_ptr())
        c.build(self.ex, kps.data_ptr(), desc.data_ptr(), n.data_ptr(), W, H, self.d_depth.data_ptr(),
                float(s["fx"]), float(s["fy"]), float(s["cx"]), float(s["cy"]), float(s["mbf"]))   # (waits, on the device, for the extraction enqueued so far: this frame's)
        self._k ^= 1
        self._extract_into(self._sets[self._k])   # the next image
        c.SearchByProjectionLast(self.last, self.table, self.th_last, mono=False, check_orientation=True, d_nmatches=self.d_nm[0].data_ptr())
        c.PoseOptimization(self.table, self.d_nm[1].data_ptr())
        c.discard_outliers()
        c.SearchLocalPoints(self.table, self.d_local.data_ptr(), self.n_local, self.th_local, self.nnratio_local, self.d_nm[2].data_ptr())
        c.PoseOptimization(self.table, self.d_nm[3].data_ptr())

    def _extract_into(self, bufs):
        B, W, H = self.B, self.W, self.H
        self.ex.extract_batch_device_async(self.d_cur.data_ptr(), B, W, H, W, W * H, bufs[0].data_ptr(), bufs[1].data_ptr(), self.cap, bufs[2].data_ptr())

    def wait_frame(self):
        """the frame of the last step_pipelined() is complete (the next image's extraction may still run)"""
        self.cur.wait()


class StereoTrackingChain(TrackingChain):
    """The same chain behind the STEREO Frame constructor (src/Frame.cc:57-113; Examples/Stereo/stereo_kitti.cc:108-117 per frame):
    both eyes' ORBextractor::operator() on a handle of their own (the two ExtractORB threads, :103-109), Frame::ComputeStereoMatches
    (:495-669) behind both on the device, then the Frame members with mvuRight / mvDepth taken from it.  The scenario is
    tracking_scenario(..., stereo=True)."""

    def __init__(self, scen: dict, device: int = 0, **kw):
        super().__init__(scen, device=device, **kw)
        t, B, cap = self.torch, self.B, self.cap
        self.ex_r = capi.Extractor(nfeatures=scen["nfeatures"], device=device)
        self.d_right = t.from_numpy(scen["right_cur"][scen["index"]]).to(self.dev)
        self.r_kps = t.zeros((B, cap, 7), dtype=t.float32, device=self.dev)
        self.r_desc = t.zeros((B, cap, 32), dtype=t.uint8, device=self.dev)
        self.r_n = t.zeros((B,), dtype=t.int32, device=self.dev)
        self.d_ur = t.zeros((B, cap), dtype=t.float32, device=self.dev)
        self.d_dp = t.zeros((B, cap), dtype=t.float32, device=self.dev)
        t.cuda.synchronize()
        self.mbf = np.float32(scen["mbf"])
        self.mb = np.float32(self.mbf / np.float32(scen["fx"]))   # mb = mbf / fx (src/Frame.cc:86)

    def enqueue_extract(self):
        B, W, H, cap = self.B, self.W, self.H, self.cap
        super().enqueue_extract()
        self.ex_r.extract_batch_device_async(self.d_right.data_ptr(), B, W, H, W, W * H, self.r_kps.data_ptr(), self.r_desc.data_ptr(), cap,
                                             self.r_n.data_ptr())
        capi.compute_stereo_matches_device_async(self.ex, self.ex_r, B, self.d_kps.data_ptr(), self.d_desc.data_ptr(), self.d_n.data_ptr(),
                                                 self.r_kps.data_ptr(), self.r_desc.data_ptr(), self.r_n.data_ptr(), cap, self.mb, self.mbf,
                                                 self.d_ur.data_ptr(), self.d_dp.data_ptr())

    def enqueue_build(self):
        W, H, s = self.W, self.H, self.scen
        self.cur.build_stereo(self.ex, self.d_kps.data_ptr(), self.d_desc.data_ptr(), self.d_n.data_ptr(), W, H, self.d_ur.data_ptr(),
                              self.d_dp.data_ptr(), float(s["fx"]), float(s["fy"]), float(s["cx"]), float(s["cy"]), float(s["mbf"]))

    def wait_extract(self):
        self.ex.wait()
        self.ex_r.wait()

    def wait(self):
        super().wait()
        self.ex_r.wait()


class ReferenceKeyFrameBoW:
    """Harness (tests/, bench.py): the front part of Tracking::TrackReferenceKeyFrame (src/Tracking.cc:858-866) for the first
    `n_frames` frames of a TrackingChain's batch, their reference keyframe being the frame's LastFrame:

        mCurrentFrame.ComputeBoW();                                        Frame::ComputeBoW, src/Frame.cc:424-431
        ORBmatcher matcher(0.7, true);
        int nmatches = matcher.SearchByBoW(mpReferenceKF, mCurrentFrame, vpMapPointMatches);

    Descriptors and keys stay in HBM: aos2_vocabulary_transform_device on the extractor's device output, ordered behind the
    extraction on the device, then aos2_matcher_search_by_bow_frames; only the FeatureVector CSRs (a few KB per frame), the
    keyframes' map point flags and the results cross PCIe.  `order()` is called by the thread that enqueued the extraction,
    `run()` may run on another thread (the calls of a step are serial per handle)."""

    def __init__(self, tc: TrackingChain, voc: dict, n_frames: int, nnratio: float = 0.7, levelsup: int = 4):
        t = tc.torch
        self.tc, self.n, self.levelsup = tc, int(n_frames), levelsup
        self.voc = capi.Vocabulary(device=tc.device)
        self.voc.set_nodes(voc["k"], voc["L"], voc["scoring"], voc["weighting"], voc["parent"], voc["desc"], voc["weight"], voc["is_leaf"])
        self.m = capi.Matcher(nnratio, True, device=tc.device)
        n, cap = self.n, tc.cap
        mk = lambda shape, dt: t.zeros(shape, dtype=dt, device=tc.dev)   # noqa: E731
        self.bw, self.bv, self.nb = mk((n, cap), t.int32), mk((n, cap), t.float64), mk((n,), t.int32)
        self.fn, self.fo, self.fi, self.nf = mk((n, cap), t.int32), mk((n, cap + 1), t.int32), mk((n, cap), t.int32), mk((n,), t.int32)
        # the reference keyframes computed their BoW when they were created (KeyFrame::ComputeBoW): once, here, into buffers
        # of their own
        self.kf = [self.bw.clone(), self.bv.clone(), self.nb.clone(), self.fn.clone(), self.fo.clone(), self.fi.clone(), self.nf.clone()]
        t.cuda.synchronize()   # torch's zero fills run on torch's stream, the transform on the vocabulary's: the fills first
        self._transform(tc.dl_desc, tc.dl_n, self.kf)
        self.kf_has_mp = np.zeros((n, cap), np.uint8)
        self.kf_has_mp[:, :] = tc.last_mp[:n] >= 0
        self.match = np.full((n, cap), -1, np.int32)
        self.nmatches = np.zeros(n, np.int32)
        t.cuda.synchronize()
        self.results, self.last_ms = None, (0.0, 0.0)

    def _transform(self, d_desc, d_n, out=None):
        o = out or [self.bw, self.bv, self.nb, self.fn, self.fo, self.fi, self.nf]
        return self.voc.transform_device(self.n, d_desc.data_ptr(), d_n.data_ptr(), self.tc.cap, self.levelsup, *(x.data_ptr() for x in o))

    def order(self):
        """the transform's stream waits for the extraction enqueued so far (device-side; the caller's thread)"""
        self.tc.ex.stream_wait(self.voc.stream())

    def run(self):
        tc = self.tc
        ms_t = self._transform(tc.d_desc, tc.d_n)    # returns when done: the extraction of this step is complete, too
        k = self.kf
        self.m.SearchByBoWFrames(self.n, tc.cap, self.kf_has_mp, self.match, self.nmatches,
                                 d_desc_kf=tc.dl_desc.data_ptr(), d_kps_kf=tc.dl_kps.data_ptr(), d_n_kf=tc.dl_n.data_ptr(),
                                 d_desc_f=tc.d_desc.data_ptr(), d_kps_f=tc.d_kps.data_ptr(), d_n_f=tc.d_n.data_ptr(),
                                 d_kf_fv_node=k[3].data_ptr(), d_kf_fv_off=k[4].data_ptr(), d_kf_fv_idx=k[5].data_ptr(), d_kf_n_fv=k[6].data_ptr(),
                                 d_f_fv_node=self.fn.data_ptr(), d_f_fv_off=self.fo.data_ptr(), d_f_fv_idx=self.fi.data_ptr(), d_f_n_fv=self.nf.data_ptr())
        self.last_ms = (ms_t, self.m.last_device_ms())
        self.results = None
        return self

    def run_calls(self):
        """the C calls of run() and nothing else (what capi.recording() captures for the native step runner)"""
        tc, k = self.tc, self.kf
        o = [self.bw, self.bv, self.nb, self.fn, self.fo, self.fi, self.nf]
        V = capi.C.c_void_p
        capi._check(self.voc.L.aos2_vocabulary_transform_device(self.voc.h, self.n, V(tc.d_desc.data_ptr()), V(tc.d_n.data_ptr()), tc.cap, self.levelsup,
                                                                *(V(x.data_ptr()) for x in o), V(None), V(None)))
        self.m.SearchByBoWFrames(self.n, tc.cap, self.kf_has_mp, self.match, self.nmatches,
                                 d_desc_kf=tc.dl_desc.data_ptr(), d_kps_kf=tc.dl_kps.data_ptr(), d_n_kf=tc.dl_n.data_ptr(),
                                 d_desc_f=tc.d_desc.data_ptr(), d_kps_f=tc.d_kps.data_ptr(), d_n_f=tc.d_n.data_ptr(),
                                 d_kf_fv_node=k[3].data_ptr(), d_kf_fv_off=k[4].data_ptr(), d_kf_fv_idx=k[5].data_ptr(), d_kf_n_fv=k[6].data_ptr(),
                                 d_f_fv_node=self.fn.data_ptr(), d_f_fv_off=self.fo.data_ptr(), d_f_fv_idx=self.fi.data_ptr(), d_f_n_fv=self.nf.data_ptr())
        self.results = None

    def get_results(self):
        """[(nmatches, match_f[:N])] of the last run (host copies)"""
        n_f = self.tc.d_n[: self.n].cpu().numpy()
        return [(int(self.nmatches[b]), self.match[b, : n_f[b]].copy()) for b in range(self.n)]


class KeyFrameWork:
    """Harness (tests/, bench.py): the matcher work of a new keyframe in LocalMapping on device-resident keyframes, for the first
    `n_kf` frames of a TrackingChain's LastFrame batch (keyframe 1 = that frame) and `n_nb` neighbour keyframes each
    (scenario.keyframe_neighbours):

        CreateNewMapPoints (src/LocalMapping.cc:272):   matcher.SearchForTriangulation(mpCurrentKeyFrame, pKF2, F12, vMatchedIndices, false)
        SearchInNeighbors  (src/LocalMapping.cc:493):   matcher.Fuse(pKFi, vpMapPointMatches)      -- the search part
        SearchInNeighbors  (src/LocalMapping.cc:518):   matcher.Fuse(mpCurrentKeyFrame, vpFuseCandidates)   -- the map points of the
                                                        neighbourhood (here: the keyframe's local map points) into the keyframe itself

    one aos2_frames_search_for_triangulation and one aos2_frames_fuse per step for all n_kf * n_nb pairs, one aos2_frames_fuse for
    the n_kf reverse problems.  Keys, descriptors,
    grids, FeatureVectors and the MapPoint table stay in HBM; the pair list, F12 and the fuse targets are host arrays, the
    results come back for the host-side map bookkeeping."""

    def __init__(self, tc: TrackingChain, voc: dict, n_kf: int, n_nb: int = 20, levelsup: int = 4, fuse_th: float = 3.0,
                 only_stereo: bool = False, check_orientation: bool = False, epipole=None, nb_cap: int | None = None, n_second: int = 0,
                 async_calls: bool = True):
        t, scen = tc.torch, tc.scen
        # n_nb first-order neighbours (CreateNewMapPoints triangulates against them, SearchInNeighbors fuses into them) and, per
        # first-order neighbour, n_second second-order ones that are fuse targets only (src/LocalMapping.cc:475-485: up to 5 each,
        # minus those that are first-order targets already)
        n_first = int(n_nb)
        self.n_first, self.n_second = n_first, int(n_second)
        n_nb = n_first * (1 + int(n_second))   # neighbour keyframes per scene = fuse targets per keyframe
        self.tc, self.n_kf, self.n_nb, self.fuse_th = tc, int(n_kf), int(n_nb), float(fuse_th)
        # CreateNewMapPoints builds `ORBmatcher matcher(0.6, false)` and passes bOnlyStereo = false (src/LocalMapping.cc:221, 272)
        self.only_stereo, self.check_orientation, self.levelsup = bool(only_stereo), bool(check_orientation), int(levelsup)
        nu, cap1, W, H = scen["n_unique"], tc.cap, tc.W, tc.H
        cap = int(nb_cap or cap1)   # keypoint capacity of the neighbour batch (the two batches of a pair may differ)
        # (only the scenes the first n_kf batch positions show need neighbour keyframes)
        self.nb = scenario.keyframe_neighbours(scen, n_nb, n_scenes=int(scen["index"][: self.n_kf].max()) + 1)
        NB = self.nb["n_scenes"] * n_nb
        self.voc = capi.Vocabulary(device=tc.device)
        self.voc.set_nodes(voc["k"], voc["L"], voc["scoring"], voc["weighting"], voc["parent"], voc["desc"], voc["weight"], voc["is_leaf"])
        # ---- the neighbour keyframes: extraction, Frame members, pose, map point flags, FeatureVectors
        self.ex = capi.Extractor(nfeatures=scen["nfeatures"], device=tc.device)
        self.d_img = t.from_numpy(self.nb["imgs"]).to(tc.dev)
        z = lambda shape, dt: t.zeros(shape, dtype=dt, device=tc.dev)   # noqa: E731
        self.n_kps, self.n_desc, self.n_n = z((NB, cap, 7), t.float32), z((NB, cap, 32), t.uint8), z((NB,), t.int32)
        t.cuda.synchronize()
        self.ex.extract_batch_device(self.d_img.data_ptr(), NB, W, H, W, W * H, self.n_kps.data_ptr(), self.n_desc.data_ptr(), cap, self.n_n.data_ptr())
        self.d_depth = t.from_numpy(np.repeat(scen["Z"][: self.nb["n_scenes"]].astype(np.float32), n_nb)).to(tc.dev)[:, None, None].expand(NB, H, W).contiguous()
        self.kfs = capi.Frames(NB, cap, tc.device)
        self.kfs.build(self.ex, self.n_kps.data_ptr(), self.n_desc.data_ptr(), self.n_n.data_ptr(), W, H, self.d_depth.data_ptr(),
                       float(scen["fx"]), float(scen["fy"]), float(scen["cx"]), float(scen["cy"]), float(scen["mbf"]))
        self.d_Tkw = t.from_numpy(np.ascontiguousarray(self.nb["Tkw"].reshape(NB, 16))).to(tc.dev)
        t.cuda.synchronize()
        self.kfs.set_pose(self.d_Tkw.data_ptr())
        rng = np.random.default_rng(66000 + scen["seed"])
        n_host = self.n_n.cpu().numpy()
        self.nb_mp = np.full((NB, cap), -1, np.int32)   # 40 % of a neighbour's features hold a map point already (row 0 stands for it)
        for j in range(NB):
            self.nb_mp[j, : n_host[j]] = np.where(rng.random(n_host[j]) < 0.4, 0, -1)
        self.kfs.set_map_points(self.nb_mp, tc.table)
        fv = lambda n: [z((n, cap), t.int32), z((n, cap), t.float64), z((n,), t.int32), z((n, cap), t.int32), z((n, cap + 1), t.int32),   # noqa: E731
                        z((n, cap), t.int32), z((n,), t.int32), z((n, cap), t.int32), z((n, cap), t.int32)]
        self.fv2 = fv(NB)
        t.cuda.synchronize()
        self.voc.transform_device(NB, self.n_desc.data_ptr(), self.n_n.data_ptr(), cap, levelsup, *(x.data_ptr() for x in self.fv2))
        # ---- keyframe 1 side: the LastFrame batch; the node of every feature (the FeatureVector key it is filed under)
        B = tc.B
        fv1 = lambda n: [z((n, cap1), t.int32), z((n, cap1), t.float64), z((n,), t.int32), z((n, cap1), t.int32), z((n, cap1 + 1), t.int32),   # noqa: E731
                         z((n, cap1), t.int32), z((n,), t.int32), z((n, cap1), t.int32), z((n, cap1), t.int32)]
        self.fv1 = fv1(B)
        t.cuda.synchronize()   # (as above: the buffers' zero fills must not land on the transform's output)
        self.voc.transform_device(B, tc.dl_desc.data_ptr(), tc.dl_n.data_ptr(), cap1, levelsup, *(x.data_ptr() for x in self.fv1))
        # ---- pairs and fuse problems
        idx = scen["index"]
        self.kf1 = np.repeat(np.arange(self.n_kf, dtype=np.int32), n_nb)
        self.kf2 = np.concatenate([idx[b] * n_nb + np.arange(n_nb) for b in range(self.n_kf)]).astype(np.int32)
        self.F12 = np.ascontiguousarray(self.nb["F12"][self.kf2])
        # (sideways motion: the epipole is at infinity; the parity sweeps pass finite ones to drive the mono-mono test of :739-745)
        self.epipole = np.zeros((len(self.kf1), 2), np.float32) if epipole is None else np.ascontiguousarray(epipole, np.float32).reshape(len(self.kf1), 2)
        # the pairs SearchForTriangulation runs on: a keyframe's first n_first neighbours; tri_of[pair] = its row in the results or -1
        self.tri_pairs = np.flatnonzero((np.arange(len(self.kf1)) % n_nb) < n_first).astype(np.int32)
        self.tri_of = np.full(len(self.kf1), -1, np.int32)
        self.tri_of[self.tri_pairs] = np.arange(len(self.tri_pairs), dtype=np.int32)
        self.t_kf1, self.t_kf2 = np.ascontiguousarray(self.kf1[self.tri_pairs]), np.ascontiguousarray(self.kf2[self.tri_pairs])
        self.t_F12, self.t_epipole = np.ascontiguousarray(self.F12[self.tri_pairs]), np.ascontiguousarray(self.epipole[self.tri_pairs])
        rows = tc.last_mp[self.kf1].copy()                          # vpMapPointMatches of keyframe 1, per (keyframe, neighbour) problem
        drop = rng.random(rows.shape) < 0.1                         # IsInKeyFrame(pKFi) / isBad(): the loop head's gate (:844-850)
        rows[drop] = -1
        self.fuse_rows = rows
        self.d_rows = t.from_numpy(rows).to(tc.dev)
        # the reverse problems (:503-518): candidates = the local map points of the keyframe's scene; the loop head (:844-850) drops
        # those the keyframe holds already (IsInKeyFrame) and a few bad ones
        loc = tc.map["local"][idx[: self.n_kf]].astype(np.int32).copy()
        for b in range(self.n_kf):
            held = tc.last_mp[b][tc.last_mp[b] >= 0]
            loc[b][np.isin(loc[b], held)] = -1
        loc[rng.random(loc.shape) < 0.05] = -1
        self.rev_rows, self.rev_target = loc, np.arange(self.n_kf, dtype=np.int32)
        self.d_rev_rows = t.from_numpy(loc).to(tc.dev)
        self.d_rev_idx, self.d_rev_dist = z(loc.shape, t.int32), z(loc.shape, t.int32)
        self.h_rev = [t.empty(loc.shape, dtype=t.int32).pin_memory() for _ in range(2)]
        P, PT = len(self.kf1), len(self.tri_pairs)
        self.d_match12, self.d_nm = z((PT, cap1), t.int32), z((PT,), t.int32)
        self.d_best_idx, self.d_best_dist = z((P, cap1), t.int32), z((P, cap1), t.int32)
        # the results of a step land in pinned host memory (one copy per array; LocalMapping's bookkeeping reads them there)
        self.h_out = [t.empty((PT, cap1), dtype=t.int32).pin_memory()] + [t.empty((P, cap1), dtype=t.int32).pin_memory() for _ in range(2)] + \
                     [t.empty((PT,), dtype=t.int32).pin_memory()]
        self.copy_stream = t.cuda.Stream(device=tc.dev)
        # the three calls of a keyframe are enqueued back to back and waited for ONCE (aos2_frames_set_async_keyframe_calls): on a device that
        # is busy with the tracking kernels every host round trip of the job costs a scheduling delay
        self.async_calls = async_calls
        tc.last.set_async_keyframe_calls(async_calls)
        self.kfs.set_async_keyframe_calls(async_calls)
        if async_calls:
            self._ext = [t.cuda.ExternalStream(tc.last.stream(), device=tc.dev), t.cuda.ExternalStream(self.kfs.stream(), device=tc.dev)]
        t.cuda.synchronize()
        self.last_ms = (0.0, 0.0)

    def run(self, timed: bool = False):
        """timed: every call waited for (last_ms = the calls' own times; bench.py's synchronous stage pass)"""
        import time
        tc = self.tc
        t0 = time.perf_counter()
        tc.last.SearchForTriangulation(self.kfs, self.t_kf1, self.t_kf2, self.t_F12, self.t_epipole, self.fv1[8].data_ptr(),
                                       [self.fv1[k].data_ptr() for k in (3, 4, 5, 6)], [self.fv2[k].data_ptr() for k in (3, 4, 5, 6)],
                                       self.d_match12.data_ptr(), self.d_nm.data_ptr(), only_stereo=self.only_stereo,
                                       check_orientation=self.check_orientation)
        if timed and self.async_calls:
            tc.last.wait()
        t1 = time.perf_counter()
        self.kfs.Fuse(tc.table, self.kf2, self.d_rows.data_ptr(), tc.cap, self.fuse_th, self.d_best_idx.data_ptr(), self.d_best_dist.data_ptr())
        tc.last.Fuse(tc.table, self.rev_target, self.d_rev_rows.data_ptr(), self.rev_rows.shape[1], self.fuse_th, self.d_rev_idx.data_ptr(),
                     self.d_rev_dist.data_ptr())
        if timed and self.async_calls:
            self.kfs.wait()
            tc.last.wait()
        t2 = time.perf_counter()
        # (synchronous calls return with their results complete; asynchronous ones are ordered before the copies on the device;
        # the copies run on a stream of this object)
        if self.async_calls:
            for e_ in self._ext:
                self.copy_stream.wait_event(e_.record_event())
        with tc.torch.cuda.stream(self.copy_stream):
            for h, d in zip(self.h_out + self.h_rev, (self.d_match12, self.d_best_idx, self.d_best_dist, self.d_nm, self.d_rev_idx, self.d_rev_dist)):
                h.copy_(d, non_blocking=True)
        self.copy_stream.synchronize()
        self.match12, self.best_idx, self.best_dist, self.nm = (h.numpy() for h in self.h_out)   # vMatchedIndices / Fuse's best_idx: host views
        self.rev_idx, self.rev_dist = (h.numpy() for h in self.h_rev)
        self.last_ms = ((t1 - t0) * 1e3, (t2 - t1) * 1e3)
        return self

    def run_calls(self):
        """the C calls of run() and nothing else (capi.recording() / the native step runner): the three searches, the results to the
        page-locked host arrays on the streams that produce them, both batches waited for"""
        tc, H = self.tc, capi.hip_runtime()
        tc.last.SearchForTriangulation(self.kfs, self.t_kf1, self.t_kf2, self.t_F12, self.t_epipole, self.fv1[8].data_ptr(),
                                       [self.fv1[k].data_ptr() for k in (3, 4, 5, 6)], [self.fv2[k].data_ptr() for k in (3, 4, 5, 6)],
                                       self.d_match12.data_ptr(), self.d_nm.data_ptr(), only_stereo=self.only_stereo,
                                       check_orientation=self.check_orientation)
        self.kfs.Fuse(tc.table, self.kf2, self.d_rows.data_ptr(), tc.cap, self.fuse_th, self.d_best_idx.data_ptr(), self.d_best_dist.data_ptr())
        tc.last.Fuse(tc.table, self.rev_target, self.d_rev_rows.data_ptr(), self.rev_rows.shape[1], self.fuse_th, self.d_rev_idx.data_ptr(),
                     self.d_rev_dist.data_ptr())
        s_last, s_kfs = tc.last.stream(), self.kfs.stream()
        for h, d, q in zip(self.h_out + self.h_rev, (self.d_match12, self.d_best_idx, self.d_best_dist, self.d_nm, self.d_rev_idx, self.d_rev_dist),
                           (s_last, s_kfs, s_kfs, s_last, s_last, s_last)):
            capi._check(H.hipMemcpyAsync(h.data_ptr(), d.data_ptr(), h.numel() * h.element_size(), capi.HIP_D2H, q))
        self.kfs.wait()
        tc.last.wait()
        self.match12, self.best_idx, self.best_dist, self.nm = (h.numpy() for h in self.h_out)   # (views of the page-locked arrays)
        self.rev_idx, self.rev_dist = (h.numpy() for h in self.h_rev)

    def snapshot(self):
        """the inputs (by reference) and the last results (copies), for oracle/parity.py"""
        import types
        return types.SimpleNamespace(kf1=self.kf1, kf2=self.kf2, nb=self.nb, nb_mp=self.nb_mp, F12=self.F12, epipole=self.epipole, tri_of=self.tri_of,
                                     fuse_rows=self.fuse_rows, n_nb=self.n_nb, fuse_th=self.fuse_th, only_stereo=self.only_stereo,
                                     check_orientation=self.check_orientation, levelsup=self.levelsup, rev_rows=self.rev_rows, rev_idx=self.rev_idx.copy(),
                                     rev_dist=self.rev_dist.copy(), match12=self.match12.copy(),
                                     nm=self.nm.copy(), best_idx=self.best_idx.copy(), best_dist=self.best_dist.copy())

"""Optional real-data inputs (harness of tests/ and bench.py): when the
environment points at a dataset -- $TUM_FR1_DESK, $KITTI_00, $EUROC_MH01 -- its frames replace the synthetic images.
No dataset is on this image or on the GPU box, so `"data": "synthetic"` is what every committed number
says; this module is what makes `"data": "real"` possible the day a box has the files.

No OpenCV, no imageio: PNG (the format of all three datasets) is decoded here with zlib + numpy, 8- and 16-bit
greyscale and 8-bit RGB / RGBA, non-interlaced.  Colour goes to grey the way the examples do it (`cv::cvtColor(...,
CV_RGB2GRAY)` in Tracking::GrabImageRGBD, src/Tracking.cc:213-226): OpenCV's 8-bit fixed-point weights
(R 4899 + G 9617 + B 1868 + 8192) >> 14.

    tum_frames(dir)     TUM RGB-D:   rgb.txt / depth.txt associated by nearest timestamp (< 0.02 s, the rule of the
                        benchmark's associate.py that produced Examples/RGB-D/associations/*.txt), depth = png / 5000
                        (DepthMapFactor of Examples/RGB-D/TUM1.yaml)
    kitti_frames(dir)   KITTI odometry sequence: image_0/%06d.png, image_1/%06d.png  (Examples/Stereo/stereo_kitti.cc)
    euroc_frames(dir)   EuRoC: mav0/cam0/data/*.png, mav0/cam1/data/*.png by name  (Examples/Stereo/stereo_euroc.cc;
                        the rectification of that example is a cv::remap outside the path and is NOT applied here)
"""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np


def read_png(path: str) -> np.ndarray:
    """-> uint8 / uint16 array [h, w] (grey) or [h, w, 3 / 4] (8-bit colour)"""
    raw = open(path, "rb").read()
    if raw[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(raw):
        (n,), kind = struct.unpack(">I", raw[pos:pos + 4]), raw[pos + 4:pos + 8]
        body = raw[pos + 8:pos + 8 + n]
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if interlace or depth not in (8, 16) or ctype not in (0, 2, 6) or (depth == 16 and ctype != 0):
        raise ValueError(f"{path}: unsupported PNG (bit depth {depth}, colour type {ctype}, interlace {interlace})")
    ch = {0: 1, 2: 3, 6: 4}[ctype]
    bpp = ch * depth // 8
    stride = w * bpp
    data = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, stride + 1)
    out = _unfilter(data, h, stride, bpp)
    if depth == 16:
        return out.reshape(h, w, 2).astype(np.uint16)[:, :, 0] * 256 + out.reshape(h, w, 2)[:, :, 1]
    return out.reshape(h, w) if ch == 1 else out.reshape(h, w, ch)


def _native_unfilter():
    """aos2_png_unfilter of the built library, or None (the loaders then fall back to numpy / Python)"""
    if os.environ.get("AOS2_PNG_PYTHON") == "1":
        return None
    try:
        import ctypes as C
        from . import capi
        fn = capi.lib().aos2_png_unfilter
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        return fn
    except Exception:   # no library / an older one: the pure-Python path below
        return None


def _unfilter(data: np.ndarray, h: int, stride: int, bpp: int) -> np.ndarray:
    """the scanline filters undone: by the library's host routine (aos2_png_unfilter) when it is built, else in numpy / Python"""
    fn = _native_unfilter()
    if fn is not None:
        buf = np.ascontiguousarray(data).copy()
        if fn(buf.ctypes.data, h, stride, bpp) != 0:
            raise ValueError("unknown PNG filter type")
        return np.ascontiguousarray(buf[:, 1:])
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft, line = int(data[y, 0]), data[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:   # Sub: a running sum per byte lane of the pixel
            cur = line.copy()
            lanes = cur.reshape(-1, bpp)
            lanes[:] = np.cumsum(lanes, axis=0) & 255
        else:           # Average / Paeth depend on the decoded left neighbour: byte by byte
            cur = line.copy()
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                if ft == 3:
                    cur[x] = (cur[x] + ((a + b) >> 1)) & 255
                else:
                    c = prev[x - bpp] if x >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    cur[x] = (cur[x] + (a if pa <= pb and pa <= pc else b if pb <= pc else c)) & 255
        out[y] = cur
        prev = cur
    return out


def write_png(path: str, img: np.ndarray):
    """8-bit grey / RGB or 16-bit grey, filter 0 (tests write small datasets with it)"""
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    if img.dtype == np.uint16:
        depth, ctype, body = 16, 0, img.astype(">u2").tobytes()
        stride = 2 * w
    else:
        depth, ctype = 8, (0 if img.ndim == 2 else 2)
        body, stride = img.astype(np.uint8).tobytes(), w * (1 if img.ndim == 2 else 3)
    rows = b"".join(b"\x00" + body[y * stride:(y + 1) * stride] for y in range(h))

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(rows, 1)) + chunk(b"IEND", b""))


def to_grey(img: np.ndarray) -> np.ndarray:
    """cv::cvtColor(CV_RGB2GRAY) on 8-bit data (fixed point, 14 fractional bits); grey passes through"""
    if img.ndim == 2:
        return img.astype(np.uint8)
    r, g, b = (img[:, :, k].astype(np.int32) for k in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def _stamped(path):
    out = []
    for ln in open(path):
        ln = ln.strip()
        if ln and not ln.startswith("#"):
            t, name = ln.replace(",", " ").split()[:2]
            out.append((float(t), name))
    return out


def tum_frames(root: str, max_difference: float = 0.02):
    """[(timestamp, rgb path, depth path)] in time order: every rgb frame with the nearest unused depth frame within 0.02 s"""
    rgb, dep = _stamped(os.path.join(root, "rgb.txt")), _stamped(os.path.join(root, "depth.txt"))
    cand = sorted((abs(a - b), a, b) for a, _ in rgb for b, _ in dep if abs(a - b) < max_difference)
    ra, rb = dict(rgb), dict(dep)
    used_a, used_b, pairs = set(), set(), []
    for _, a, b in cand:
        if a not in used_a and b not in used_b:
            used_a.add(a)
            used_b.add(b)
            pairs.append((a, os.path.join(root, ra[a]), os.path.join(root, rb[b])))
    return sorted(pairs)


def kitti_frames(root: str):
    left = os.path.join(root, "image_0")
    names = sorted(n for n in os.listdir(left) if n.endswith(".png"))
    return [(i, os.path.join(left, n), os.path.join(root, "image_1", n)) for i, n in enumerate(names)]


def euroc_frames(root: str):
    left = os.path.join(root, "mav0", "cam0", "data")
    names = sorted(n for n in os.listdir(left) if n.endswith(".png"))
    return [(int(n[:-4]), os.path.join(left, n), os.path.join(root, "mav0", "cam1", "data", n)) for n in names
            if os.path.exists(os.path.join(root, "mav0", "cam1", "data", n))]


def dataset_from_env(cfg: str):
    """(kind, directory) if the environment variable of the configuration names an existing directory, else None"""
    var = {"tum": "TUM_FR1_DESK", "kitti": "KITTI_00", "euroc": "EUROC_MH01"}[cfg]
    d = os.environ.get(var)
    return (cfg, d) if d and os.path.isdir(d) else None


def tum_pairs(root: str, n_pairs: int, step: int = 1, depth_factor: float = 5000.0):
    """n_pairs (LastFrame, CurrentFrame) pairs of consecutive associated frames: grey images uint8 [n, h, w] and depth maps in
    metres float32 [n, h, w] (imDepth.convertTo(CV_32F, 1 / DepthMapFactor), src/Tracking.cc:226-227; 0 = no measurement)"""
    fr = tum_frames(root)
    if len(fr) < 2:
        raise ValueError(f"{root}: fewer than two associated frames")
    idx = [(i * step) % (len(fr) - 1) for i in range(n_pairs)]
    load = lambda k: (to_grey(read_png(fr[k][1])), (read_png(fr[k][2]).astype(np.float32) * np.float32(1.0 / depth_factor)).astype(np.float32))   # noqa: E731
    last, cur = [load(i) for i in idx], [load(i + 1) for i in idx]
    return dict(last=np.stack([a for a, _ in last]), cur=np.stack([a for a, _ in cur]), depth_last=np.stack([d for _, d in last]),
                depth_cur=np.stack([d for _, d in cur]), stamps=[fr[i][0] for i in idx])

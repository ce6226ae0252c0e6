"""The optional real-data path (active-orb-slam2_amd/datasets.py): PNG decoding without OpenCV (every filter type, 8- / 16-bit
grey, RGB), cv::cvtColor's RGB -> grey weights, the TUM timestamp association, and the scenario built from recorded pairs.
The datasets themselves are absent here and on the GPU box: the tests write a miniature TUM directory."""
import os
import struct
import zlib

import numpy as np
import pytest


def _png_with_filters(path, img):
    """encoder of the TEST: row y uses filter type y % 5, so that the decoder's Sub / Up / Average / Paeth paths all run"""
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    if img.dtype == np.uint16:
        depth, ctype, bpp = 16, 0, 2
        raw = np.frombuffer(img.astype(">u2").tobytes(), np.uint8).reshape(h, w * 2).astype(np.int32)
    else:
        depth, ctype, bpp = 8, (0 if img.ndim == 2 else 2), (1 if img.ndim == 2 else 3)
        raw = img.reshape(h, -1).astype(np.int32)
    rows = bytearray()
    prev = np.zeros(raw.shape[1], np.int32)
    for y in range(h):
        cur, ft = raw[y], y % 5
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - a
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((a + prev) >> 1)
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            f = cur - np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        rows += bytes([ft]) + (f & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(bytes(rows))[:40]) + chunk(b"IDAT", zlib.compress(bytes(rows))[40:]) + chunk(b"IEND", b""))


@pytest.mark.parametrize("native", [True, False])
def test_png_decoder_all_filters(pkg, tmp_path, monkeypatch, native):
    D = pkg.datasets
    if not native:
        monkeypatch.setenv("AOS2_PNG_PYTHON", "1")   # the numpy / Python fallback instead of the library's host routine
    assert (D._native_unfilter() is not None) == native
    rng = np.random.default_rng(0)
    for name, img in (("g8", rng.integers(0, 256, (37, 53), dtype=np.uint8)), ("g16", rng.integers(0, 65536, (21, 30)).astype(np.uint16)),
                      ("rgb", rng.integers(0, 256, (19, 41, 3), dtype=np.uint8))):
        _png_with_filters(tmp_path / (name + ".png"), img)
        assert np.array_equal(D.read_png(str(tmp_path / (name + ".png"))), img), name
        D.write_png(str(tmp_path / (name + "_w.png")), img)
        assert np.array_equal(D.read_png(str(tmp_path / (name + "_w.png"))), img), name
    with pytest.raises(ValueError):
        (tmp_path / "x.png").write_bytes(b"not a png")
        D.read_png(str(tmp_path / "x.png"))
    # cv::cvtColor(CV_RGB2GRAY), 8-bit: pure colours and a grey ramp
    rgb = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [10, 10, 10], [0, 0, 0]]], np.uint8)
    assert D.to_grey(rgb).tolist() == [[76, 150, 29, 255, 10, 0]]


def _mini_tum(pkg, root, n=5):
    D = pkg.datasets
    os.makedirs(root / "rgb")
    os.makedirs(root / "depth")
    canvas = pkg.synth.synth_image(77, 700, 520)
    rgb_lines, dep_lines = ["# color images", "# file: mini", "# timestamp filename"], ["# depth maps", "# x", "# timestamp filename"]
    for i in range(n):
        t = 1305031452.0 + i / 30.0
        g = canvas[10 + i:490 + i, 20 + 2 * i:660 + 2 * i]
        D.write_png(str(root / "rgb" / f"{t:.6f}.png"), np.stack([g, g, g], 2))   # grey as RGB: to_grey returns g
        d = np.full((480, 640), 10000, np.uint16)   # 2 m at the factor 5000
        d[100:140, 200:260] = 0
        D.write_png(str(root / "depth" / f"{t + 0.004:.6f}.png"), d)
        rgb_lines.append(f"{t:.6f} rgb/{t:.6f}.png")
        dep_lines.append(f"{t + 0.004:.6f} depth/{t + 0.004:.6f}.png")
    dep_lines.append(f"{1305031452.0 + 99:.6f} depth/none.png")   # a depth frame without an rgb partner
    (root / "rgb.txt").write_text("\n".join(rgb_lines) + "\n")
    (root / "depth.txt").write_text("\n".join(dep_lines) + "\n")
    return canvas


def test_tum_association_and_real_scenario(pkg, tmp_path, monkeypatch):
    D = pkg.datasets
    canvas = _mini_tum(pkg, tmp_path)
    fr = D.tum_frames(str(tmp_path))
    assert len(fr) == 5 and all(abs(float(os.path.basename(b)[:-4]) - t - 0.004) < 1e-4 for t, _, b in fr) and fr == sorted(fr)
    assert D.dataset_from_env("tum") is None
    monkeypatch.setenv("TUM_FR1_DESK", str(tmp_path))
    assert D.dataset_from_env("tum") == ("tum", str(tmp_path))
    pr = D.tum_pairs(str(tmp_path), 3)
    assert pr["last"].shape == (3, 480, 640) and pr["last"].dtype == np.uint8 and pr["depth_last"].dtype == np.float32
    assert np.array_equal(pr["last"][1], canvas[11:491, 22:662]) and np.array_equal(pr["cur"][1], canvas[12:492, 24:664])
    assert pr["depth_last"][0, 0, 0] == np.float32(2.0) and pr["depth_last"][0, 120, 230] == 0
    scen = pkg.scenario.tracking_scenario_real(pr, 6)
    assert scen["batch"] == 6 and scen["n_unique"] == 3 and (scen["index"] == [0, 1, 2, 0, 1, 2]).all() and scen["real"]
    assert np.array_equal(scen["Tlw"][0], np.eye(4, dtype=np.float32))

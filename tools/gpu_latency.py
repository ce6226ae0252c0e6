import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
img = pkg.synth.synth_image(5)
ex = pkg.Extractor()
ex(img)
ts = []
for i in range(30):
    t = time.perf_counter(); ex(img); ts.append(time.perf_counter() - t)
print("single frame host API: median %.3f ms" % (np.median(ts) * 1e3), ex.last_timing())
for B in (16, 256):
    imgs = pkg.synth.synth_batch(0, min(B, 32)); imgs = np.concatenate([imgs] * (B // len(imgs)))
    ex.extract_batch(imgs)
    t = time.perf_counter()
    for i in range(5): ex.extract_batch(imgs)
    dt = (time.perf_counter() - t) / 5
    print("host API batch %d, pageable input: %.2f ms -> %.0f frames/s (python unpack of results included)" % (B, dt * 1e3, B / dt))
    pin = pkg.host_empty(imgs.shape, np.uint8); pin[...] = imgs
    ex.extract_batch(pin)
    t = time.perf_counter()
    for i in range(5): ex.extract_batch(pin)
    dt = (time.perf_counter() - t) / 5
    print("host API batch %d, page-locked input: %.2f ms -> %.0f frames/s (python unpack included)" % (B, dt * 1e3, B / dt))
    # the C call alone (what a C++ caller sees): page-locked in and out, no numpy unpacking
    import ctypes as C
    h, w = imgs.shape[1:]; cap = ex.max_keypoints_for(w, h)
    kps = pkg.host_empty((B, cap), pkg.capi.KP_DTYPE); desc = pkg.host_empty((B, cap, 32), np.uint8); n = np.zeros(B, np.int32)
    for src, name in ((imgs, "pageable"), (pin, "page-locked")):
        ts = []
        for i in range(7):
            t = time.perf_counter()
            rc = ex.L.aos2_extractor_extract_batch(ex.h, src.ctypes.data_as(C.c_void_p), B, w, h, w, w * h, kps.ctypes.data_as(C.c_void_p),
                                                   desc.ctypes.data_as(C.c_void_p), cap, n.ctypes.data_as(C.c_void_p))
            assert rc == 0
            ts.append(time.perf_counter() - t)
        dt = float(np.median(ts))
        print("  C call only, %s input, page-locked output: %.2f ms -> %.0f frames/s  (%.1f GB/s over PCIe)" % (
            name, dt * 1e3, B / dt, (src.nbytes + kps.nbytes + desc.nbytes) / dt / 1e9))

"""One-off parity sweep (a time-boxed slice of it runs in tests/test_fuzz_gpu.py): random image sizes / extractor parameters, HIP path vs oracle,
keypoints + descriptors + every pyramid level compared bit for bit.  python tools/gpu_fuzz_extractor.py [n_cases [seconds]]"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 1e18   # optional time budget in seconds (tests/test_fuzz_gpu.py)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(12345)
bad = 0
rejected = 0
t0 = time.time()
ran = 0
for c in range(n_cases):
    if time.time() - t0 > budget_s:
        break
    ran += 1
    w = int(rng.integers(96, 1300)); h = int(rng.integers(96, 800))
    h = min(h, int(1.9 * w))   # round(w / h) == 0 is undefined behaviour in the reference (division by zero, :545)
    nf = int(rng.choice([100, 500, 1000, 2000, 3000]))
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.35, 1.5, 2.0]))
    nl = int(rng.integers(1, 9))
    ini = int(rng.choice([20, 20, 12, 40])); mn = int(rng.choice([7, 7, 5, ini]))
    # every level must keep a FAST-able interior
    while nl > 1 and min(w, h) / sf ** (nl - 1) < 60: nl -= 1
    img = pkg.synth.synth_image(1000 + c, w, h)
    if c % 5 == 4: img = np.clip(img.astype(np.int32) // 3 + 90, 0, 255).astype(np.uint8)   # low contrast: minThFAST path
    try:
        ex = pkg.Extractor(nfeatures=nf, scale_factor=sf, nlevels=nl, ini_th=ini, min_th=mn)
        oe = O.Extractor(nfeatures=nf, scale_factor=sf, nlevels=nl, ini_th=ini, min_th=mn)
    except Exception as e:
        print(c, "ctor", (w, h, nf, sf, nl, ini, mn), repr(e)); continue
    try:
        k, d = ex(img)
    except Exception as e:
        if "too small" in repr(e) or "twice as tall" in repr(e):    # documented rejections (the reference itself fails there)
            rejected += 1
        else:
            bad += 1; print('ERROR case', c, (w, h, nf, sf, nl, ini, mn), repr(e))
        continue
    ok_, od = oe.extract(img)
    same = len(k) == len(ok_) and (d == od).all() and all((k[f] == ok_[f]).all() for f in k.dtype.names)
    pyr_ok = True
    for l in range(nl):
        a = ex.pyramid_level(l)
        b = oe.level_plane(l)
        if a.shape != b.shape or not (a == b).all(): pyr_ok = False
    if not (same and pyr_ok):
        bad += 1
        print("MISMATCH case", c, (w, h, nf, sf, nl, ini, mn), "n", len(k), len(ok_), "pyr_ok", pyr_ok)
    # a batch of the same geometry through the batched path: three images (the per-job octree kernel with helper waves) or nine (two levels
    # per workgroup: octree_pair_kernel), alternately
    if c % 4 == 0:
        imgs = np.stack([img, img[::-1].copy(), np.roll(img, 7, axis=1)] + ([] if c % 8 else [np.roll(img, k, axis=(k & 1)) for k in (3, 8, 13, 18, 23, 28)]))
        res = ex.extract_batch(imgs)
        for im, (kb, db) in zip(imgs, res):
            k1, d1 = oe.extract(im)
            if len(kb) != len(k1) or not (db == d1).all():
                bad += 1; print("BATCH MISMATCH case", c, (w, h, nf, sf, nl))
print("cases", ran, "of", n_cases, "rejected by design", rejected, "MISMATCHES / ERRORS", bad, "time %.1f s" % (time.time() - t0))

"""Stage-by-stage parity dump of the HIP extractor vs the oracle (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
cfgs = [("tum", 640, 480, 1000), ("kitti", 1241, 376, 2000), ("small", 320, 240, 300)]
for name, w, h, nf in cfgs:
    img = pkg.synth.synth_image(1, w, h)
    ex = pkg.Extractor(nfeatures=nf); oe = O.Extractor(nfeatures=nf)
    t = time.time(); kps, desc = ex(img); t1 = time.time() - t
    okps, odesc = oe.extract(img)
    print(name, "n", len(kps), len(okps), "first call %.1f ms" % (t1 * 1e3), ex.last_timing())
    for l in range(8):
        a = ex.pyramid_level(l); b = oe.level_plane(l)
        x, y, s = ex.debug_candidates(l); ox, oy, os_ = oe.level_candidates(l)
        same = len(x) == len(ox) and (x == ox).all() and (y == oy).all() and (s == os_).all()
        print("  L%d pyr_equal=%s ncand=%d/%d cand_equal=%s" % (l, a.shape == b.shape and (a == b).all(), len(x), len(ox), same))
        if not same and len(x) and len(ox):
            so = set(zip(ox.tolist(), oy.tolist(), os_.tolist())); sg = set(zip(x.tolist(), y.tolist(), s.tolist()))
            print("     only_oracle", sorted(so - sg)[:5], "only_gpu", sorted(sg - so)[:5])
    if len(kps) == len(okps):
        for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
            print("  field", f, "equal:", (kps[f] == okps[f]).all(), "nbad", int((kps[f] != okps[f]).sum()))
        print("  desc equal:", (desc == odesc).all(), "rows bad", int((desc != odesc).any(axis=1).sum()))
    t = time.time(); kps2, _ = ex(img); print("  second call %.2f ms" % ((time.time() - t) * 1e3), ex.last_timing())
    bdr = ex.pyramid_level(2, border=19)
    print("  border ok:", (bdr == O.copy_make_border(oe.level_plane(2))).all())
# batch + timing
imgs = pkg.synth.synth_batch(100, 16)
ex = pkg.Extractor()
res = ex.extract_batch(imgs)
oe = O.Extractor()
bad = 0
for b in range(16):
    okps, odesc = oe.extract(imgs[b])
    k, d = res[b]
    if len(k) != len(okps) or (k.tobytes() != okps.tobytes()) or (d != odesc).any():
        bad += 1
print("batch16 mismatching images:", bad, ex.last_timing())
res = ex.extract_batch(imgs); print("batch16 again", ex.last_timing(), "fast kernel ms", ex.bench_fast(10), "describe ms", ex.bench_describe(10))
a = np.linspace(0, 6.2832, 100000).astype(np.float32)
s, c = pkg.capi.debug_sincos_device(a)
hs = np.array([pkg.capi.debug_sincos_host(v) for v in a[:20000]], np.float32)
print("sincos device==host:", (s[:20000] == hs[:, 0]).all() and (c[:20000] == hs[:, 1]).all())
for B in (64, 256):
    imgs = pkg.synth.synth_batch(1000, B)
    ex = pkg.Extractor()
    res = ex.extract_batch(imgs)
    t = time.time(); res = ex.extract_batch(imgs); dt = time.time() - t
    print("batch", B, "wall %.1f ms -> %.0f fps" % (dt * 1e3, B / dt), ex.last_timing(), "fast ms", ex.bench_fast(10), "describe ms", ex.bench_describe(10))
os.environ["AOS2_OCTREE"] = "host"
ex = pkg.Extractor()
res = ex.extract_batch(imgs)
t = time.time(); res2 = ex.extract_batch(imgs); dt = time.time() - t
print("host-octree batch", len(imgs), "wall %.1f ms -> %.0f fps" % (dt * 1e3, len(imgs) / dt), ex.last_timing(), "nproc", os.cpu_count())
print("host==device octree:", all((a[0].tobytes() == b[0].tobytes()) and (a[1] == b[1]).all() for a, b in zip(res, res2)))

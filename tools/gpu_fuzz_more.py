"""One-off parity sweep (a time-boxed slice of it runs in tests/test_fuzz_gpu.py) of the order-independent searches, the vocabulary and the asynchronous
extractor call with awkward batch sizes, vs the oracle.  python tools/gpu_fuzz_more.py [n_cases [seconds]]"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle(); S = pkg.synth
budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 1e18   # optional time budget in seconds (tests/test_fuzz_gpu.py)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(2468)
bad = 0
t0 = time.time()


def flag(name, c, params):
    global bad
    bad += 1
    print("MISMATCH", name, "case", c, params, flush=True)


dev = torch.device("cuda:0")
ran = 0
for c in range(n_cases):
    if time.time() - t0 > budget_s:
        break
    ran += 1
    ori = bool(rng.integers(0, 2))
    # SearchForTriangulation
    n1, n2, nn = int(rng.choice([30, 600, 2100])), int(rng.choice([25, 700, 1900])), int(rng.choice([1, 20, 150]))
    kw = [dict(), dict(only_stereo=True), dict(mono=True), dict(cfg="tum")][c % 4]
    p = S.synth_triang_problem(3000 + c, n1, n2, n_nodes=nn, check_orientation=ori, **kw)
    a, b = pkg.Matcher(0.6, ori).SearchForTriangulation(p, only_stereo=kw.get("only_stereo", False)), O.search_for_triangulation(p)
    if a[0] != b[0] or not (a[1] == b[1]).all(): flag("triang", c, (n1, n2, nn, kw))
    # Fuse (both overloads) and SearchBySim3
    nf, npts = int(rng.choice([40, 900, 2500])), int(rng.choice([10, 1500, 4000]))
    f, pg = S.synth_proj_gen_problem(3100 + c, n_f=nf, n_pts=npts, cfg=("kitti", "tum", "euroc")[c % 3], th=float(rng.choice([2.5, 3.0, 4.0])),
                                     stereo=bool(c % 3))
    for sim3 in (False, True):
        a, b = pkg.Matcher().Fuse(f, pg, sim3=sim3), O.fuse(f, pg, sim3=sim3)
        if a[0] != b[0] or not (a[1] == b[1]).all() or not (a[2] == b[2]).all(): flag("fuse", c, (nf, npts, sim3))
    f1, f2, p12, p21 = S.synth_sim3_problem(3200 + c, int(rng.choice([50, 900, 2000])), int(rng.choice([60, 1000, 1800])), cfg=("kitti", "euroc")[c % 2])
    a, b = pkg.Matcher().SearchBySim3(f1, f2, p12, p21), O.search_by_sim3(f1, f2, p12, p21)
    if a[0] != b[0] or not (a[1] == b[1]).all(): flag("sim3", c, ())
    # SearchForInitialization
    f2i, q = S.synth_init_problem(3300 + c, int(rng.choice([50, 1200, 2500])), int(rng.choice([60, 1500, 2400])))
    ws, ratio = int(rng.choice([10, 30, 100])), float(rng.choice([0.7, 0.9]))
    a, b = pkg.Matcher(ratio, ori).SearchForInitialization(f2i, q, ws), O.search_for_initialization(f2i, q, ws, ratio, ori)
    if a[0] != b[0] or not (a[1] == b[1]).all(): flag("init", c, (ws, ratio, ori))
    # ComputeDistinctiveDescriptors
    off, desc = S.synth_observations(3400 + c, int(rng.choice([1, 300, 2500])), int(rng.choice([2, 24, 90])))
    if not (pkg.Matcher().ComputeDistinctiveDescriptors(off, desc) == O.compute_distinctive_descriptors(off, desc)).all(): flag("distinctive", c, ())
    # vocabulary transform
    if c % 2 == 0:
        k, L = int(rng.choice([2, 5, 10, 13])), int(rng.choice([1, 2, 3, 5]))
        voc = S.synth_vocabulary(3500 + c, k, L, ragged=bool(c % 4 == 2))
        V, OV = pkg.Vocabulary(), O.Vocabulary()
        for X in (V, OV): X.set_nodes(voc["k"], voc["L"], voc["scoring"], voc["weighting"], voc["parent"], voc["desc"], voc["weight"], voc["is_leaf"])
        d = S.vocab_descriptors(np.random.default_rng(c), voc, int(rng.choice([1, 50, 1000, 3000])))
        lu = int(rng.integers(0, L + 2))
        ra, rb = V.transform(d, lu), OV.transform(d, lu)
        if any(ra[key].tobytes() != rb[key].tobytes() for key in rb): flag("vocab", c, (k, L, lu, len(d)))
    # asynchronous extractor with awkward batch sizes (chunk boundaries, grids padded to 8 images)
    if c % 4 == 0:
        B = int(rng.choice([1, 3, 7, 9, 63, 65, 97, 130]))
        w, h = int(rng.choice([320, 640, 752])), int(rng.choice([240, 480]))
        imgs = S.synth_batch(3600 + c, min(B, 5), w, h)
        imgs = np.concatenate([imgs] * (-(-B // len(imgs))))[:B]
        ex = pkg.Extractor(nfeatures=int(rng.choice([300, 1000])))
        cap = ex.max_keypoints_for(w, h)
        d_img = torch.from_numpy(imgs).to(dev)
        outs = [(torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
                 torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(2)]
        for o in outs:
            ex.extract_batch_device_async(d_img.data_ptr(), B, w, h, w, w * h, o[0].data_ptr(), o[1].data_ptr(), cap, o[2].data_ptr())
        ex.wait(); torch.cuda.synchronize()
        oe = O.Extractor(nfeatures=ex.nfeatures)
        want = [oe.extract(im) for im in imgs[: min(B, 5)]]
        for o in outs:
            n = o[2].cpu().numpy(); kk = o[0].cpu().numpy().view(pkg.capi.KP_DTYPE).reshape(B, cap); dd = o[1].cpu().numpy()
            for bi in range(B):
                wk, wd = want[bi % len(want)]
                if n[bi] != len(wk) or kk[bi, : n[bi]].tobytes() != wk.tobytes() or not (dd[bi, : n[bi]] == wd).all():
                    flag("async", c, (B, w, h, bi)); break
print("cases", ran, "of", n_cases, "MISMATCHES", bad, "time %.1f s" % (time.time() - t0))

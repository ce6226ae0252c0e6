"""What the parity conventions of DESIGN.md section 2 are worth (CPU only, the oracle against itself): each convention that a reference
binary could resolve differently -- the octree's order of equal-size nodes (heap addresses there), libm cosf / sinf in the rBRIEF
steering, libm logf in MapPoint::PredictScale, the order of a landmark's edges in LocalBundleAdjustment (pointer order of the
observations there) -- is flipped in the ORACLE, and the outputs are compared with the default convention's: how many keypoints,
descriptor bits, matches, float32 steps of a pose change.  The numbers say how large the differences of a pinned run (the reference
built with its own OpenCV / Eigen beside this oracle) can be expected to be per convention, before any real disagreement.

    python tools/convention_effects.py > profiles/r05_convention_effects.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as g

pkg = g.load_package()
O = g.load_oracle()
L = O.lib()
S = pkg.synth
NF = int(os.environ.get("CONV_FRAMES", "32"))
t0 = time.time()
print("# Effects of the parity conventions (DESIGN.md section 2), the oracle with a convention flipped against the oracle as tested; %d synthetic 640x480 frames" % NF)
print("# (tools/convention_effects.py; CPU only)")
imgs = S.synth_batch(20_000, NF, 640, 480)


def extract_all():
    ex = O.Extractor(nfeatures=1000)
    return [ex.extract(im) for im in imgs]


def best_matches(d1, d2):
    """index of the Hamming-nearest row of d2 for every row of d1 (ties: lowest index)"""
    x = np.unpackbits(d1, axis=1).astype(np.int16)
    y = np.unpackbits(d2, axis=1).astype(np.int16)
    dist = x @ (1 - y).T + (1 - x) @ y.T
    return dist.argmin(1), dist.min(1)


base = extract_all()
nk = sum(len(k) for k, _ in base)

# ---- 1. rBRIEF steering: libm cosf / sinf instead of the correctly rounded pair
L.orc_set_trig_mode(1)
trig = extract_all()
L.orc_set_trig_mode(0)
kp_diff = sum(int(a[0].tobytes() != b[0].tobytes()) for a, b in zip(base, trig))
rows = sum(int((a[1] != b[1]).any(1).sum()) for a, b in zip(base, trig))
bits = sum(int(np.unpackbits(a[1] ^ b[1]).sum()) for a, b in zip(base, trig))
mchg = 0
for i in range(0, NF - 1, 2):   # nearest-neighbour matches between consecutive frames, both conventions
    m0, _ = best_matches(base[i][1], base[i + 1][1])
    m1, _ = best_matches(trig[i][1], trig[i + 1][1])
    mchg += int((m0 != m1).sum())
print("\n1. cos / sin of the steering angle (ORBextractor.cc:112-113): glibc cosf / sinf against the correctly rounded float pair")
print("   frames whose keypoints change: %d of %d (the angle itself is not affected)" % (kp_diff, NF))
print("   descriptors with a changed bit: %d of %d (%.3f %%), bits changed: %d of %d (%.2e of all bits; %.2f per frame)" %
      (rows, nk, 100.0 * rows / nk, bits, nk * 256, bits / (nk * 256.0), bits / NF))
print("   nearest-neighbour matches (frame i -> frame i + 1, %d frame pairs) that change: %d of %d" % (NF // 2, mchg, sum(len(base[i][0]) for i in range(0, NF - 1, 2))))

# ---- 2. octree tie-break: equal-size nodes in the opposite order
L.orc_set_tiebreak_mode(1)
tie = extract_all()
L.orc_set_tiebreak_mode(0)
fr = 0
moved = 0
for (k0, _), (k1, _) in zip(base, tie):
    s0 = set(map(bytes, k0.view(np.uint8).reshape(len(k0), -1)))
    s1 = set(map(bytes, k1.view(np.uint8).reshape(len(k1), -1)))
    d = len(s0 - s1)
    moved += d
    fr += int(d > 0 or len(k0) != len(k1))
print("\n2. DistributeOctTree's order of equal-size nodes (ORBextractor.cc:684 sorts pair<int, ExtractorNode*>: ties by heap address): creation order against its reverse")
print("   frames whose keypoint SET changes: %d of %d; keypoints of the default result that are not in the flipped one: %d of %d (%.3f %%; %.1f per frame)" %
      (fr, NF, moved, nk, 100.0 * moved / nk, moved / NF))
print("   (which nodes are divided last when the target count is reached inside a group of equal-size nodes decides which cells keep one keypoint and which four)")

# ---- 3. PredictScale's logarithm: libm logf against the correctly rounded float
rng = np.random.default_rng(5)
ratios = np.exp(rng.uniform(0.0, np.log(1.2 ** 8), 2_000_000)).astype(np.float32)
lg_cr = np.log(ratios.astype(np.float64)).astype(np.float32)
import ctypes as C
libm = C.CDLL("libm.so.6")
libm.logf.restype = C.c_float
libm.logf.argtypes = [C.c_float]
lg_m = np.array([libm.logf(float(r)) for r in ratios[:200_000]], np.float32)
lsf = np.float32(np.log(np.float32(1.2)))
lev_cr = np.ceil(lg_cr[:200_000] / lsf)
lev_m = np.ceil(lg_m / lsf)
print("\n3. log in MapPoint::PredictScale (src/MapPoint.cc:435): glibc logf against (float)log((double)ratio)")
print("   of 200 000 random distance ratios in [1, 1.2^8]: logf differs in the last bit for %d (%.3f %%), the predicted level differs for %d" %
      (int((lg_m != lg_cr[:200_000]).sum()), 100.0 * float((lg_m != lg_cr[:200_000]).mean()), int((lev_cr != lev_m).sum())))
chg = tot = 0
for seed in range(6):
    f, mp = S.synth_proj_mp_problem(50 + seed)
    a = O.search_by_projection_mp(f, mp)
    L.orc_set_log_mode(1)
    b = O.search_by_projection_mp(f, mp)
    L.orc_set_log_mode(0)
    chg += int((np.asarray(a[1]) != np.asarray(b[1])).sum())
    tot += len(a[1])
print("   SearchByProjection(F, vpMapPoints) on 6 synthetic problems: matches that change: %d of %d features" % (chg, tot))

# ---- 4. LocalBundleAdjustment: the order of the edges (the reference walks every map point's observations in pointer order)
print("\n4. order of the edges handed to the optimiser (Optimizer.cc:583-586 iterates a std::map keyed by KeyFrame*): reversed against ascending mnId")
for name, prob in [("SURVEY 8(d) window", S.synth_lba_problem(0, n_points=8000))] + [("mix window %d" % i, S._lba_from_kwargs(m)) for i, m in enumerate(S.lba_window_mix(0, 3))]:
    w0 = O.lba_solve(prob)
    q = dict(prob)
    order = np.arange(prob["n_edges"])[::-1].copy()
    for k in ("edge_pose", "edge_point", "edge_obs", "edge_stereo", "edge_inv_sigma2"):
        q[k] = np.ascontiguousarray(prob[k][order])
    w1 = O.lba_solve(q)
    dp = np.abs(w0["pose_Tcw"] - w1["pose_Tcw"])
    dx = np.abs(w0["point_xyz"] - w1["point_xyz"])
    ulp = lambda a, b: int(np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)).max())   # noqa: E731
    d64 = float(np.abs(w0["pose_qt"] - w1["pose_qt"]).max()), float(np.abs(w0["point_xyz64"] - w1["point_xyz64"]).max())
    print("   %-20s %5d edges: iterations %s / %s, trials %d / %d, outlier flags that differ %d; float32 poses: max |diff| %.2e (%d float32 steps), %d of %d entries differ; "
          "float32 points: max |diff| %.2e, %d of %d entries differ; in double before the write-back: poses %.1e, points %.1e" %
          (name, prob["n_edges"], w0["iters"], w1["iters"], w0["trials"], w1["trials"], int((w0["edge_outlier"] != w1["edge_outlier"][np.argsort(order)]).sum()),
           float(dp.max()), ulp(w0["pose_Tcw"], w1["pose_Tcw"]), int((dp > 0).sum()), dp.size, float(dx.max()), int((dx > 0).sum()), dx.size, d64[0], d64[1]))

print("\n# %.0f s" % (time.time() - t0))

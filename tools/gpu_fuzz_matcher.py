"""One-off parity sweep of the greedy / projection searches (a time-boxed slice of it runs in tests/test_fuzz_gpu.py): random problem sizes, densities,
radii, ratios and orientation flags, HIP path (parallel fixed-point stage B) vs the oracle.
python tools/gpu_fuzz_matcher.py [n_cases [seconds]]"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle(); S = pkg.synth
budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 1e18   # optional time budget in seconds (tests/test_fuzz_gpu.py)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(777)
bad = 0
t0 = time.time()


def chk(name, c, params, got, want):
    global bad
    ok = got[0] == want[0] and (np.asarray(got[1]) == np.asarray(want[1])).all()
    if not ok:
        bad += 1
        print("MISMATCH", name, "case", c, params, "n", got[0], want[0])


ran = 0
for c in range(n_cases):
    if time.time() - t0 > budget_s:
        break
    ran += 1
    nf = int(rng.choice([8, 60, 300, 1000, 2000, 3100, 8000]))
    nq = int(rng.choice([5, 200, 1500, 3072, 3073, 4500]))
    th = float(rng.choice([1.0, 3.0, 7.0, 15.0, 40.0]))
    ratio = float(rng.choice([0.6, 0.75, 0.9, 0.95])); ori = bool(rng.integers(0, 2))
    if nf >= 8000: nq = min(nq, 1500)
    # SearchByProjection(F, vpMapPoints)
    f, mp = S.synth_proj_mp_problem(5000 + c, n_f=nf, n_mp=nq, th=min(th, 15.0), nnratio=ratio)
    if c % 3 == 0: mp["has_obs"] = (np.arange(nq) % 2).astype(np.uint8)
    chk("proj_mp", c, (nf, nq, th, ratio), pkg.Matcher(ratio, True).SearchByProjection(f, mp, th=float(mp["th"])), O.search_by_projection_mp(f, mp))
    # SearchByProjection(Current, Last)
    n = int(rng.choice([50, 700, 1500, 3200]))
    cur, pl = S.synth_proj_last_problem(6000 + c, n=n, th=float(rng.choice([7.0, 15.0])), mono=bool(c % 4 == 1), check_orientation=ori)
    chk("proj_last", c, (n,), pkg.Matcher(0.9, ori).SearchByProjectionLast(cur, pl, float(pl["th"]), int(pl["mono"])), O.search_by_projection_last(cur, pl))
    # SearchByBoW (KF, F) and (KF, KF)
    nk, nff, nn = int(rng.choice([40, 500, 1500, 2500])), int(rng.choice([30, 400, 1200, 2600])), int(rng.choice([1, 3, 40, 200]))
    p = S.synth_bow_problem(7000 + c, nk, nff, n_nodes=nn, nnratio=ratio, check_orientation=ori)
    chk("bow", c, (nk, nff, nn, ratio, ori), pkg.Matcher(ratio, ori).SearchByBoW(p), O.search_by_bow(p))
    p = S.synth_bow_kf_problem(8000 + c, nk, nff, n_nodes=nn, nnratio=ratio, check_orientation=ori)
    chk("bow_kf", c, (nk, nff, nn, ratio, ori), pkg.Matcher(ratio, ori).SearchByBoWKF(p), O.search_by_bow_kf(p))
    # SearchByProjection(pKF, Scw) and the relocalisation search
    fg, pg = S.synth_proj_gen_problem(9000 + c, n_f=min(nf, 3100), n_pts=nq, cfg=("kitti", "tum")[c % 2], th=float(rng.choice([4, 10, 25])))
    chk("proj_kf", c, (fg["n_f"], nq), pkg.Matcher().SearchByProjectionKF(fg, pg), O.search_by_projection_kf(fg, pg))
    od = int(rng.choice([64, 100]))
    chk("reloc", c, (fg["n_f"], nq, od, ori), pkg.Matcher(0.9, ori).SearchByProjectionReloc(fg, pg, od), O.search_by_projection_reloc(fg, pg, od, ori))
    # batched SearchByProjection == single
    if c % 5 == 0:
        probs = [S.synth_proj_mp_problem(5500 + 10 * c + s, n_f=int(rng.choice([60, 900, 2000])), n_mp=int(rng.choice([100, 1300, 3100])), th=3.0) for s in range(5)]
        res = pkg.Matcher(0.8, True).SearchByProjectionBatch([q[0] for q in probs], [q[1] for q in probs], th=3.0)
        for (ff, mm), r in zip(probs, res):
            chk("proj_batch", c, (ff["n_f"], mm["n_mp"]), r, O.search_by_projection_mp(ff, mm))
print("cases", ran, "of", n_cases, "MISMATCHES", bad, "time %.1f s" % (time.time() - t0))

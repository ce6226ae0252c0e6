"""Matcher + LocalBA parity dump vs the oracle (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
S = pkg.synth
rng = np.random.default_rng(5)
m = pkg.Matcher(0.7, True)
q = S.synth_descriptors(rng, 2000); t = S.synth_descriptors(rng, 2000)
t[:500] = S.flip_bits(rng, q[300:800], 0.05)
bi, bd, sd = m.hamming_best2(q, t)
D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(axis=2)
obi = D.argmin(axis=1); obd = D.min(axis=1); Ds = np.sort(D, axis=1)
print("hamming best idx/dist/second equal:", (bi == obi).all(), (bd == obd).all(), (sd == Ds[:, 1]).all())
for seed in range(4):
    p = S.synth_bow_problem(seed, 1000 + 200 * seed, 1000, nnratio=0.7)
    n0, m0 = O.search_by_bow(p)
    n1, m1 = m.SearchByBoW(p)
    print("bow seed", seed, "nmatches", n0, n1, "equal", n0 == n1 and (m0 == m1).all())
probs = [S.synth_bow_problem(10 + i, 1000, 1000, nnratio=0.7) for i in range(64)]
t0 = time.time(); res = m.SearchByBoW(probs); dt = time.time() - t0
ok = all((O.search_by_bow(p)[0] == r[0]) and (O.search_by_bow(p)[1] == r[1]).all() for p, r in zip(probs, res))
print("bow batch64 equal:", ok, "wall %.1f ms" % (dt * 1e3))
m2 = pkg.Matcher(0.8, True)
for seed in range(4):
    f, mp = S.synth_proj_mp_problem(seed)
    n0, m0 = O.search_by_projection_mp(f, mp)
    n1, m1 = m2.SearchByProjection(f, {k: v for k, v in mp.items() if k not in ("th", "nnratio")}, th=float(mp["th"]))
    print("proj_mp seed", seed, n0, n1, "equal", n0 == n1 and (m0 == m1).all())
m3 = pkg.Matcher(0.9, True)
for seed in range(6):
    cur, p = S.synth_proj_last_problem(seed, mono=(seed == 5))
    n0, m0 = O.search_by_projection_last(cur, p)
    pp = {k: v for k, v in p.items() if k not in ("th", "mono", "check_orientation")}
    n1, m1 = m3.SearchByProjectionLast(cur, pp, float(p["th"]), int(p["mono"]))
    print("proj_last seed", seed, n0, n1, "equal", n0 == n1 and (m0 == m1).all(), "culled", int((m0 == -2).sum()))
ba = pkg.LocalBA()
for cfgp in (dict(seed=1, n_local=3, n_fixed=2, n_points=60, stereo_frac=0.5), dict(seed=2, n_local=6, n_fixed=4, n_points=400, stereo_frac=0.0),
             dict(seed=0), dict(seed=3, include_kf0=True)):
    prob = S.synth_lba_problem(**cfgp)
    t0 = time.time(); ro = O.lba_solve(prob); to = time.time() - t0
    rg = ba.LocalBundleAdjustment(prob)
    t0 = time.time(); rg = ba.LocalBundleAdjustment(prob); tg = time.time() - t0
    dp = np.abs(rg["pose_Tcw"] - ro["pose_Tcw"]).max(); dx = np.abs(rg["point_xyz"] - ro["point_xyz"]).max()
    print("lba", cfgp, "edges", prob["n_edges"], "iters", ro["iters"], rg["iters"], "max|dT|=%.3g max|dX|=%.3g" % (dp, dx),
          "outlier_equal", (rg["edge_outlier"] == ro["edge_outlier"]).all(), "chi2 %.6f vs %.6f" % (rg["final_chi2"], ro["chi2_trace"][-1]),
          "cpu %.1f ms gpu wall %.1f ms dev %.1f ms" % (to * 1e3, tg * 1e3, rg["ms_device"]))

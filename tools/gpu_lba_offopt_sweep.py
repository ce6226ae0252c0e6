"""LocalBundleAdjustment windows that START far from the optimum, many of them: the device against the oracle beside the oracle's own
resolution on each window (oracle/parity.py lba_resolution) -- how often does the rule `max(1e-5, LBA_RESOLUTION_FACTOR x the oracle's spread)` hold, and
by what margin?      python tools/gpu_lba_offopt_sweep.py [n_windows] > profiles/r06_lba_offopt_sweep.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
import parity
from concurrent.futures import ThreadPoolExecutor
S = pkg.synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
rng = np.random.default_rng(7)
mix = S.lba_window_mix(5, N, hard_every=1)
levels = [(0.5, 3.0, 2.0), (0.3, 1.0, 0.5), (1.0, 6.0, 4.0), (0.2, 0.5, 0.2)]
for i, m in enumerate(mix):
    m["hard"] = levels[i % len(levels)]
    m["n_points"] = 800 + m["n_points"] // 6     # (smaller windows: the oracle runs seven times per window)
probs = S.synth_lba_problems(mix)
t0 = time.time()
ba = pkg.LocalBA()
got = ba.LocalBundleAdjustmentBatch(probs)
slots, rounds = ba.last_program()
def ref(p):
    w = O.lba_solve(p)
    return w, parity.lba_resolution(p, want=w)
with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as pool:
    refs = list(pool.map(ref, probs))
rows, bad, dec_bad, unconv, detail = [], 0, 0, 0, []
for i, (p, r, (w, res)) in enumerate(zip(probs, got, refs)):
    dp = float(np.abs(r["pose_Tcw"] - w["pose_Tcw"]).max()); dx = float(np.abs(r["point_xyz"] - w["point_xyz"]).max())
    same = tuple(r["iters"]) == tuple(w["iters"]) and sum(r["trials"]) == w["trials"] and bool((r["edge_outlier"] == w["edge_outlier"]).all())
    mm = parity.lba_mismatches(r, w, tag="window %d" % i, resolution=res)
    bad += bool(mm); dec_bad += not same
    rows.append((i, mix[i]["hard"], w["iters"], w["trials"], dp, res["pose"], dx, res["point"], same, res["decisions_equal"], len(mm)))
    if not same:
        detail.append("#   window %d: device iters %s trials %s, oracle iters %s trials %d, outlier flags that differ %d of %d; final chi2 device %.10e oracle %.10e lambda %.6e / %.6e; oracle variants: %s" %
                      (i, tuple(r["iters"]), tuple(r["trials"]), tuple(w["iters"]), w["trials"], int((r["edge_outlier"] != w["edge_outlier"]).sum()), len(w["edge_outlier"]),
                       r["final_chi2"], w["chi2_trace"][-1], r["final_lambda"], w["lambda_trace"][-1],
                       "; ".join("%s: %s" % (x["name"][:1], "=" if x["decisions_equal"] else "%d flags" % x["outlier_flags_differ"]) for x in res["rows"])))
print("# %d windows started off the optimum (four perturbation levels), one device batch (%d trial slots, %d host rounds); %.0f s" % (N, slots, rounds, time.time() - t0))
print("# window  perturbation  oracle iters / trials | poses: device-vs-oracle  oracle-vs-itself | points: device-vs-oracle  oracle-vs-itself | decisions equal (device, oracle variants) | rule")
for r in rows:
    print("%4d  %-16s %-8s %3d | %.2e  %.2e | %.2e  %.2e | %s %s | %s" % (r[0], r[1], tuple(r[2]), r[3], r[4], r[5], r[6], r[7], r[8], r[9], "ok" if r[10] == 0 else "EXCEEDED"))
dp = np.array([r[4] for r in rows]); sp = np.array([r[5] for r in rows]); dx = np.array([r[6] for r in rows]); sx = np.array([r[7] for r in rows])
print("\n".join(detail))
vals_bad = sum(1 for (i, p, r, (w, res)) in [(i, probs[i], got[i], refs[i]) for i in range(N)]
               if float(np.abs(r["pose_Tcw"] - w["pose_Tcw"]).max()) > max(1e-5, parity.LBA_RESOLUTION_FACTOR * res["pose"]) or float(np.abs(r["point_xyz"] - w["point_xyz"]).max()) > max(1e-5, parity.LBA_RESOLUTION_FACTOR * res["point"]))
outl_bad = sum(1 for i in range(N) if not (got[i]["edge_outlier"] == refs[i][0]["edge_outlier"]).all())
print("# poses / points beyond max(1e-5, %g x the oracle's own spread on the window)" % parity.LBA_RESOLUTION_FACTOR + ": %d of %d windows; outlier SETS that differ: %d; iteration or trial COUNTS that differ: %d "
      "(listed above: windows whose second optimisation ends early on a final chi2 of 0 .. 1e-27 or a handful of active edges -- the count hinges on an exact floating-point zero in "
      "rho, levenberg.cpp:118-150 -- and on most of which the oracle's own re-associated runs do not agree with each other either); oracle variants disagreeing among themselves: %d windows" %
      (vals_bad, N, outl_bad, dec_bad, sum(1 for r in rows if not r[9])))
print("# device-vs-oracle above 1e-5: poses %d windows (worst %.2e), points %d windows (worst %.2e); oracle-vs-itself above 1e-5: poses %d (worst %.2e), points %d (worst %.2e)" %
      (int((dp > 1e-5).sum()), dp.max(), int((dx > 1e-5).sum()), dx.max(), int((sp > 1e-5).sum()), sp.max(), int((sx > 1e-5).sum()), sx.max()))
ratio = np.maximum(dp / np.maximum(1e-5, sp), dx / np.maximum(1e-5, sx))
print("# device difference / max(1e-5, oracle spread), the larger of poses and points: median %.2f, 90 %% %.2f, max %.2f (the rule allows %g)" %
      (np.median(ratio), np.quantile(ratio, 0.9), ratio.max(), parity.LBA_RESOLUTION_FACTOR))

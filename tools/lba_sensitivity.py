"""LocalBundleAdjustment on windows that start FAR from the optimum (synth.perturb_lba_problem): how far do two faithful executions
of the SAME algorithm end apart?  The oracle is run against itself (CPU only; oracle/parity.py lba_resolution) with its arithmetic
re-associated at rounding level:
  (a) the edges handed over in the opposite order (Optimizer.cc:583-586 walks a std::map keyed by KeyFrame*: pointer order),
  (b) the reduced system eliminated last-unknown-first (another ordering of linear_solver_eigen.h:94-124's factorisation),
  (c) every accumulation across edges / landmarks / pivots carried in long double and rounded once,
  (d) = (b) + (c),
  (e) the same C restatement compiled with the flags the reference's own build gives g2o (Thirdparty/g2o/CMakeLists.txt:57,
      -O3 -march=native: gcc then fuses a * b + c; oracle/Makefile "contracted") -- every product-sum rounds once instead of twice,
      INSIDE an edge's error, Jacobians and blocks too, which (a)-(d) leave untouched (a landmark with two inlier edges has no
      order to change),
  (f) = (e) + (b).
Each variant computes every quantity to within a few units of the last place of the default run's; whatever separates the END
results is the optimisation's own amplification of such differences over 15 Levenberg-Marquardt iterations that did not converge.
The spread per window is the resolution with which ANY implementation (the device path, a build of the reference with another
Eigen version, compiler or -march) can be compared with the oracle on that window.

    python tools/lba_sensitivity.py > profiles/r06_lba_sensitivity.txt        (LBA_SENS_WINDOWS=n: the bench mix of n windows)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as g

pkg = g.load_package()
O = g.load_oracle()
import parity  # noqa: E402  (oracle/parity.py)

S = pkg.synth


def hard_test_problems():
    """the windows tests/test_lba_gpu.py starts off the optimum (its _hard_problem and the hard ones of its mix)"""
    def hp(seed, rot, tr, pt):
        prob = S.synth_lba_problem(seed, n_local=5, n_fixed=3, n_points=250, stereo_frac=0.3)
        return S.perturb_lba_problem(prob, seed, rot, tr, pt)
    out = [("test _hard_problem(42, 0.5, 3, 2)", hp(42, 0.5, 3.0, 2.0)), ("test _hard_problem(43, 1.0, 6, 4)", hp(43, 1.0, 6.0, 4.0))]
    mix = S.lba_window_mix(11, 12, hard_every=3)
    for m in mix:
        m["n_points"] = 900 + m["n_points"] // 8
    for i, m in enumerate(mix):
        out.append(("test mix(11, 12, hard_every=3) window %d%s" % (i, " HARD" if "hard" in m else ""), S._lba_from_kwargs(m)))
    return out


def bench_problems(n):
    mix = S.lba_window_mix(0, n, hard_every=8)
    return [("bench mix(0, %d, hard_every=8) window %d%s" % (n, i, " HARD" if "hard" in m else ""), S._lba_from_kwargs(m))
            for i, m in enumerate(mix) if "hard" in m or i % 8 == 0]


if __name__ == "__main__":
    t0 = time.time()
    print("# LocalBundleAdjustment, the oracle against itself with its arithmetic re-associated (tools/lba_sensitivity.py; CPU only)")
    print("# per window: iterations / trials of the default run, then per variant max |diff| of the float32 write-back (16 entries of Tcw per")
    print("# keyframe; points), the same in double before the write-back, and whether every decision (iterations, trials, outlier set) is the default run's")
    worst = {}
    for name, prob in hard_test_problems() + bench_problems(int(os.environ.get("LBA_SENS_WINDOWS", "64"))):
        w0 = O.lba_solve(prob)
        res = parity.lba_resolution(prob, want=w0)
        hard = "HARD" in name or "_hard_problem" in name
        print("\n%s: %d free keyframes, %d edges, iterations %s, trials %d, chi2 %.4g -> %.4g" %
              (name, int((prob["pose_fixed"] == 0).sum()), prob["n_edges"], tuple(w0["iters"]), w0["trials"], w0["chi2_trace"][0], w0["chi2_trace"][-1]))
        for r in res["rows"]:
            print("   %-48s poses %.2e  points %.2e  (double: %.1e  %.1e)  decisions %s" %
                  (r["name"], r["pose"], r["point"], r["pose64"], r["point64"], "equal" if r["decisions_equal"] else "DIFFER (%d outlier flags)" % r["outlier_flags_differ"]))
        w = worst.setdefault("off the optimum" if hard else "ordinary", [0.0, 0.0, 0, 0])
        w[0], w[1], w[2], w[3] = max(w[0], res["pose"]), max(w[1], res["point"]), w[2] + (0 if res["decisions_equal"] else 1), w[3] + 1
    print()
    for k, w in worst.items():
        print("# %d windows %s: largest spread poses %.2e, points %.2e; windows with a run whose decisions differ: %d" % (w[3], k, w[0], w[1], w[2]))
    print("# %.0f s" % (time.time() - t0))

"""rocprofv3 target: repeated single-problem greedy searches (stage A + fixed-point stage B kernels)"""
import sys
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.synth
f, mp = S.synth_proj_mp_problem(3, n_f=1000, n_mp=1500, th=3.0)
m = pkg.Matcher(float(mp["nnratio"]), True)
for _ in range(20):
    m.SearchByProjection(f, mp, th=float(mp["th"]))
fk, p = S.synth_proj_gen_problem(20, n_f=2000, n_pts=3000, cfg="tum", th=10)
for _ in range(20):
    pkg.Matcher().SearchByProjectionKF(fk, p)

import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); S = pkg.synth
m = pkg.Matcher(0.7, True)
probs = [S.synth_bow_problem(100 + i, 1000, 1000, nnratio=0.7) for i in range(64)]
for _ in range(3):
    t = time.perf_counter(); m.SearchByBoW(probs); print("bow64 wall ms", (time.perf_counter() - t) * 1e3)
for _ in range(3):
    t = time.perf_counter(); m.SearchByBoW(probs[0]); print("bow1 wall ms", (time.perf_counter() - t) * 1e3)
f, mp = S.synth_proj_mp_problem(0)
m2 = pkg.Matcher(0.8, True)
for _ in range(3):
    t = time.perf_counter(); m2.SearchByProjection(f, mp, th=3.0); print("proj_mp wall ms", (time.perf_counter() - t) * 1e3)
cur, p = S.synth_proj_last_problem(0)
m3 = pkg.Matcher(0.9, True)
for _ in range(3):
    t = time.perf_counter(); m3.SearchByProjectionLast(cur, p, 7.0, 0); print("proj_last wall ms", (time.perf_counter() - t) * 1e3)

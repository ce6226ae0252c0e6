import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g, numpy as np, importlib.util
pkg=g.load_package(); O=g.load_oracle()
spec=importlib.util.spec_from_file_location('t','tests/test_lba_gpu.py'); t=importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
prob=t._hard_problem(pkg,43,0.5,3,2)
w=O.lba_solve(prob)
print(w['iters'], w['trials'], w['chi2_trace'], w['lambda_trace'], int(w['edge_level1'].sum()), prob['n_edges'])
r=pkg.LocalBA().LocalBundleAdjustment(prob)
print(r['iters'], r['trials'])

import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
ba = pkg.LocalBA()
u = [pkg.synth.synth_lba_problem(i, n_points=8000) for i in range(4)]
probs = [u[i % 4] for i in range(32)]
prep = ba.prepare_batch(probs)
ba.solve_prepared(prep)
for _ in range(3):
    t0 = time.time(); ba.solve_prepared(prep); print("wall %.2f ms dev %.2f" % ((time.time() - t0) * 1e3, prep["R"][0].ms_device), file=sys.stderr)

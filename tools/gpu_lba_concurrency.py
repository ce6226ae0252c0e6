# LocalBA batches in flight: throughput of K concurrent aos2_lba_solve_batch calls (own handle + host thread each)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import __graft_entry__ as g
pkg = g.load_package()
u = [pkg.synth.synth_lba_problem(i, n_points=8000) for i in range(4)]
nwin = int(os.environ.get("NWIN", "32"))
probs = [u[i % 4] for i in range(nwin)]
for K in (1, 2, 4):
    bas = [pkg.LocalBA() for _ in range(K)]
    preps = [b.prepare_batch(probs) for b in bas]
    for b, p in zip(bas, preps):
        b.solve_prepared(p)
    pool = ThreadPoolExecutor(K)
    R = 6
    t0 = time.time()
    jobs = [pool.submit(lambda b=b, p=p: [b.solve_prepared(p) for _ in range(R)]) for b, p in zip(bas, preps)]
    [j.result() for j in jobs]
    dt = time.time() - t0
    print("handles %d windows/batch %d: %.2f ms per batch (device of one: %.2f ms)" % (K, nwin, dt * 1e3 / (R * K), preps[0]["R"][0].ms_device))

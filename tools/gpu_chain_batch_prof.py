"""Device-resident tracking chain at the bench batch size (256 frames per step), nothing else on the device: for
rocprofv3 --kernel-trace --stats (tools/prof_chain_batch.sh)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
B = int(os.environ.get("CHAIN_B", "256"))
scen = pkg.scenario.tracking_scenario(5, B, n_unique=min(B, 32))
tc = pkg.chain.TrackingChain(scen, n_local=1500)
for _ in range(3):
    tc.step(); tc.wait()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 10
for _ in range(N):
    tc.step(); tc.wait()
print("chain step (B=%d, synchronous): %.3f ms" % (B, (time.perf_counter() - t0) * 1e3 / N), file=sys.stderr)

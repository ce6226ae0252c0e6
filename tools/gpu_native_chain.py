"""Single-sequence chain (B = 1) with its C calls issued by native code (csrc/host_runner.cpp replaying the recorded calls of
TrackingChain.step / wait: what a C++ caller of the C ABI does) against the same calls from Python: how much of the plain chain's
0.68 ms per frame is the interpreter's enqueue rate."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
capi = pkg.capi
scen = pkg.scenario.tracking_scenario(5, 1, n_unique=1)
tc = pkg.chain.TrackingChain(scen, n_local=1500)
for _ in range(20):
    tc.step(); tc.wait()
ref = tc.d_nm.cpu().numpy().ravel().copy()
N = 400
a = time.perf_counter()
for _ in range(N):
    tc.step(); tc.wait()
print("python calls %.4f ms per frame" % ((time.perf_counter() - a) / N * 1e3))
r = capi.Runner(1, 1)
with capi.recording() as c:
    tc.cur.wait()
r.set_list(r.PIPE_WAIT, 0, c)
with capi.recording() as c:
    tc.step()
r.set_list(r.PIPE_STEP, 0, c)
r.run(0, 20); r.sync()
tc.d_nm.zero_(); torch.cuda.synchronize()
a = time.perf_counter()
r.run(0, N); r.sync()
print("native calls %.4f ms per frame (enqueue + wait per frame, %d calls per frame)" % ((time.perf_counter() - a) / N * 1e3, len(c)))
got = tc.d_nm.cpu().numpy().ravel()
print("matches equal", bool((got == ref).all()))

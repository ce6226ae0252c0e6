"""200 frames of chain.step_pipelined (B = 1) for a kernel trace: which kernels of the next image's extraction run beside the tracking
kernels of this frame (tools/call: rocprofv3 --kernel-trace, then the last frames' timeline by queue)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
scen = pkg.scenario.tracking_scenario(5, 1, n_unique=1)
extra = int(os.environ.get("EXTRA_STREAMS", "0"))   # streams created before the chain's (moves the chain's streams to other hardware queues)
keep = [torch.cuda.Stream() for _ in range(extra)]
for s_ in keep:
    with torch.cuda.stream(s_):
        torch.zeros(4, device="cuda")
tc = pkg.chain.TrackingChain(scen, n_local=1500)
for _ in range(20):
    tc.step(); tc.wait()
a = time.perf_counter()
for _ in range(100):
    tc.step(); tc.wait()
print("plain     %.4f ms per frame" % ((time.perf_counter() - a) * 10))
for _ in range(10):
    tc.step_pipelined(); tc.wait_frame()
a = time.perf_counter()
for _ in range(200):
    tc.step_pipelined(); tc.wait_frame()
print("pipelined %.4f ms per frame" % ((time.perf_counter() - a) * 5))
tc.wait()

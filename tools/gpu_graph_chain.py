"""Single-sequence chain (B = 1) replayed as one hipGraph: the calls of TrackingChain.step() are captured once from the Frame batch's
stream (the extractor's stream joins through aos2_extractor_wait_for_stream and leaves through the wait Frame::build makes), then one
hipGraphLaunch per frame.  Compares wall per frame against the plain call sequence; checks the match counts."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
capi = pkg.capi
scen = pkg.scenario.tracking_scenario(5, 1, n_unique=1)
tc = pkg.chain.TrackingChain(scen, n_local=1500)
H = capi.hip_runtime()
vp = C.c_void_p
for _ in range(20):
    tc.step(); tc.wait()
ref = tc.d_nm.cpu().numpy().ravel().copy()
def timeit(fn, n=200):
    for _ in range(10): fn()
    a = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - a) / n * 1e3
def plain():
    tc.step(); tc.wait()
print("plain  %.4f ms per frame" % timeit(plain))
s_f = tc.cur.stream()
mode = int(os.environ.get("CAPTURE_MODE", "2"))   # 0 global, 1 thread local, 2 relaxed
st = H.hipStreamBeginCapture(vp(s_f), mode)
print("begin capture:", st)
tc.ex.wait_for_stream(s_f)
tc.step()
graph = vp()
st = H.hipStreamEndCapture(vp(s_f), C.byref(graph))
print("end capture:", st, graph.value)
if st != 0:
    sys.exit(1)
n_nodes = C.c_size_t(0)
H.hipGraphGetNodes(graph, None, C.byref(n_nodes))
print("nodes:", n_nodes.value)
ge = vp()
st = H.hipGraphInstantiate(C.byref(ge), graph, None, None, 0)
print("instantiate:", st)
tc.d_nm.zero_(); torch.cuda.synchronize()
def replay():
    H.hipGraphLaunch(ge, vp(s_f)); H.hipStreamSynchronize(vp(s_f))
print("graph  %.4f ms per frame" % timeit(replay))
got = tc.d_nm.cpu().numpy().ravel()
print("matches", got, "reference", ref, "equal", bool((got == ref).all()))

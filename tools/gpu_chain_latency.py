"""Single-frame latency of the device-resident tracking chain (B = 1): the reference's real calling pattern
(Tracking::Track: one frame at a time).  Wall clock of enqueue + wait, median of 50."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
scen = pkg.scenario.tracking_scenario(5, 1, n_unique=1)
tc = pkg.chain.TrackingChain(scen, n_local=1500)
B, W, H, cap, s = 1, tc.W, tc.H, tc.cap, scen
c = tc.cur
args = (float(s["fx"]), float(s["fy"]), float(s["cx"]), float(s["cy"]), float(s["mbf"]))
def t(fn, n=50):
    v = []
    for _ in range(n + 5):
        torch.cuda.synchronize(); a = time.perf_counter(); fn(); v.append(time.perf_counter() - a)
    return float(np.median(v[5:])) * 1e3
def extract():
    tc.ex.extract_batch_device_async(tc.d_cur.data_ptr(), B, W, H, W, W * H, tc.d_kps.data_ptr(), tc.d_desc.data_ptr(), cap, tc.d_n.data_ptr()); tc.ex.wait()
def motion():
    c.set_pose(tc.d_guess.data_ptr())
    tc.ex.extract_batch_device_async(tc.d_cur.data_ptr(), B, W, H, W, W * H, tc.d_kps.data_ptr(), tc.d_desc.data_ptr(), cap, tc.d_n.data_ptr())
    c.build(tc.ex, tc.d_kps.data_ptr(), tc.d_desc.data_ptr(), tc.d_n.data_ptr(), W, H, tc.d_depth.data_ptr(), *args)
    c.SearchByProjectionLast(tc.last, tc.table, tc.th_last, False, True, tc.d_nm[0].data_ptr())
    c.PoseOptimization(tc.table, tc.d_nm[1].data_ptr())
    c.wait()
def full():
    tc.step(); tc.wait()
r = {"extract_ms": t(extract), "extract+frame+search_last+pose_ms": t(motion), "full_chain_ms": t(full)}
# (the pipelined sequence -- next image beside this frame -- is measured and traced by tools/gpu_pipelined_trace.py)
def stage(fn):
    def f():
        fn(); c.wait()
    return f
extract()
r["frame_build_ms"] = t(stage(lambda: (c.build(tc.ex, tc.d_kps.data_ptr(), tc.d_desc.data_ptr(), tc.d_n.data_ptr(), W, H, tc.d_depth.data_ptr(), *args), c.set_pose(tc.d_guess.data_ptr()))))
r["search_last_ms"] = t(stage(lambda: c.SearchByProjectionLast(tc.last, tc.table, tc.th_last, False, True, tc.d_nm[0].data_ptr())))
r["pose_opt_ms"] = t(stage(lambda: c.PoseOptimization(tc.table, tc.d_nm[1].data_ptr())))
r["search_local_ms"] = t(stage(lambda: c.SearchLocalPoints(tc.table, tc.d_local.data_ptr(), tc.n_local, tc.th_local, tc.nnratio_local, tc.d_nm[2].data_ptr())))
print(r)
print("matches", tc.d_nm.cpu().numpy().ravel())

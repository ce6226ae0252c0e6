import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
scen = pkg.scenario.tracking_scenario(9, 3, n_unique=3)
scen["cur"][1][:] = 0          # a blank current frame: no keypoints
scen["cur"][2][:, :] = scen["cur"][2][0, 0]   # constant
tc = pkg.chain.TrackingChain(scen, n_local=800)
tc.step(); tc.wait()
F = pkg.capi.Frames
print("n", tc.d_n.cpu().numpy(), "nm", tc.d_nm.cpu().numpy())
T = tc.cur.get(F.TCW)
print("pose kept for empty frames:", [bool(np.array_equal(T[b], scen["Tcw_guess"][b].reshape(16))) for b in range(3)])
print("mp all -1:", [(tc.cur.get(F.MAP_POINTS)[b] == -1).all() for b in range(3)])

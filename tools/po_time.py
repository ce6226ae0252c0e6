"""Phase cycle counters of the frame batch's PoseOptimization at B = 1 (library built by tools/build_po_timing_lib.sh, AOS2_LIB=...):
the kernel's workgroup 0 prints them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
scen = pkg.scenario.tracking_scenario(5, 1, n_unique=1)
tc = pkg.chain.TrackingChain(scen, n_local=1500)
for _ in range(3):
    tc.step(); tc.wait()

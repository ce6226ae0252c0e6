"""One window with many free keyframes (reduced system beyond LDS: k_ldlt_dev), phase cycle counters of the reduced-system kernel
(build with tools/build_ldlt_timing_lib.sh, run with AOS2_LIB=.../libaos2_ldlttiming.so AOS2_LBA_TRACE=1)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
nl = int(os.environ.get("N_LOCAL", "40"))
p = pkg.synth.synth_lba_problem(7, n_local=nl, n_fixed=20, n_points=6000)
ba = pkg.LocalBA()
ba.LocalBundleAdjustment(p)
r = ba.LocalBundleAdjustment(p)
print("n_local", nl, "edges", p["n_edges"], "device ms", r["ms_device"], "trials", r["trials"], file=sys.stderr)

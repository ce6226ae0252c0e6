"""phase cycle counters of pose_optimization_kernel for several problem sizes (needs lib/libaos2_potiming.so via AOS2_LIB)"""
import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g, numpy as np
pkg = g.load_package()
ba = pkg.LocalBA()
for n in (60, 250, 500, 750, 1000):
    p = pkg.synth.synth_pose_problem(4000 + n, n=n, stereo_frac=0.8, outlier_frac=0.1, cfg="tum")
    print("n", n, flush=True)
    r = ba.PoseOptimization(p)
    r = ba.PoseOptimization(p)
    print(" device ms", ba.last_pose_device_ms() if hasattr(ba, "last_pose_device_ms") else None, flush=True)

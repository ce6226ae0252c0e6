"""Phase breakdown of the level-0 octree jobs (needs lib built with -DAOS2_OCT_PROF; AOS2_LIB=.../libaos2_prof.so)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
B = int(os.environ.get("OCT_B", "256"))
imgs = np.concatenate([pkg.synth.synth_batch(0, 32)] * 8)[:B]
ex = pkg.Extractor()
ex.set_chunks(1)
ex.extract_batch(imgs)
L = pkg.capi.lib()
out = (C.c_longlong * 16)()
L.aos2_debug_oct_prof(out, 1)
ex.extract_batch(imgs)
L.aos2_debug_oct_prof(out, 1)
v = np.array(out[:], dtype=np.float64)
jobs = v[12]
print("timing", {k: round(x, 3) for k, x in ex.last_timing().items()})
print("level-0 jobs", jobs)
names = ["roots+bucket", "main passes", "sort", "final divides", "best response", "list walk"]
for i, nm in enumerate(names):
    print(f"  {nm:14s} {v[i] / jobs / 100.0:8.2f} us/job")   # wall_clock64 = 100 MHz
print(f"  main divides/job {v[8]/jobs:.1f}  passes/job {v[9]/jobs:.2f}  sorted pairs/job {v[10]/jobs:.1f}  final divides/job {v[11]/jobs:.1f}  nodes/job {v[13]/jobs:.1f}  walk steps/job {v[14]/jobs:.1f}")

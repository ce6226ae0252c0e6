"""ORBextractor::operator() alone over B resident frames (640x480; `kitti` as second argument: 1241x376, 2000 features), one stream (AOS2_CHUNKS=1), for rocprofv3 passes:
    rocprofv3 --kernel-trace --stats ... -- python tools/prof_extract.py 512 [kitti]
    rocprofv3 --kernel-trace --pmc FETCH_SIZE ... (WRITE_SIZE, SQ_* in passes of their own)
B = 512: pyramids + candidate slots of a batch = 2 x the 256 MiB Infinity Cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AOS2_CHUNKS", "1")
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
CFG = pkg.synth.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "tum"]   # "kitti": 1241 x 376, 2000 features (one eye per image)
W, H = CFG["w"], CFG["h"]
base = pkg.synth.synth_batch(10_000, 32, W, H)
d = torch.from_numpy(np.concatenate([base] * (B // 32 + 1))[:B]).cuda()
ex = pkg.Extractor(nfeatures=CFG["nfeatures"])
cap = ex.max_keypoints
k = torch.empty((B, cap, 7), dtype=torch.float32, device="cuda")
ds = torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda")
n = torch.empty((B,), dtype=torch.int32, device="cuda")
for _ in range(6):
    ex.extract_batch_device(d.data_ptr(), B, W, H, W, W * H, k.data_ptr(), ds.data_ptr(), cap, n.data_ptr())
print("B", B, "keypoints/frame", float(n.float().mean()), "stage_ms", ex.last_timing(), "fast_ms", ex.bench_fast(10), file=sys.stderr)

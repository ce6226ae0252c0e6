import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import __graft_entry__ as g
pkg = g.load_package()
B = 256
imgs = np.concatenate([pkg.synth.synth_batch(0, 32)] * 8)
d = torch.from_numpy(imgs).cuda()
ex = pkg.Extractor(); cap = ex.max_keypoints
k = torch.empty((B, cap, 7), dtype=torch.float32, device="cuda"); ds = torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda"); n = torch.empty(B, dtype=torch.int32, device="cuda")
def run(c, iters=20):
    ex.set_chunks(c)
    for i in range(3): ex.extract_batch_device(d.data_ptr(), B, 640, 480, 640, 640 * 480, k.data_ptr(), ds.data_ptr(), cap, n.data_ptr())
    t = time.perf_counter()
    for i in range(iters): ex.extract_batch_device(d.data_ptr(), B, 640, 480, 640, 640 * 480, k.data_ptr(), ds.data_ptr(), cap, n.data_ptr())
    return (time.perf_counter() - t) / iters * 1e3
for rep in range(2):
    for c in (1, 2, 3, 4, 1):
        ms = run(c)
        print("rep", rep, "chunks", c, "ms/step %.3f" % ms, {k_: round(v, 3) for k_, v in ex.last_timing().items()} if c == 1 else "", "fast_ms %.3f" % ex.bench_fast(10))

import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
W, H = 640, 480
uni = np.stack([pkg.synth.synth_image(300 + i, W, H) for i in range(16)])
oe = O.Extractor(nfeatures=1000)
want = [oe.extract(u) for u in uni]
for B in (720, 1920):
    for chunks in (0, 1):
        d_img = torch.from_numpy(uni[np.arange(B) % 16]).cuda()
        ex = pkg.Extractor(nfeatures=1000)
        ex.set_chunks(chunks)
        cap = ex.max_keypoints_for(W, H)
        kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
        n = torch.zeros((B,), dtype=torch.int32, device="cuda")
        ex.extract_batch_device(d_img.data_ptr(), B, W, H, W, W * H, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr())
        nh = n.cpu().numpy(); dh = desc.cpu().numpy()
        bad = [b for b in range(B) if nh[b] != len(want[b % 16][0]) or not (dh[b, :nh[b]] == want[b % 16][1]).all()]
        print("B", B, "chunks", chunks, "bad images", len(bad), bad[:8], bad[-3:])

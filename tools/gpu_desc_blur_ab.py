"""A/B of the descriptor stage: per-keypoint blur inside describe_kernel (default) against the reference's own form -- a
whole-level GaussianBlur pass + describe on the blurred planes (AOS2_DESC_BLUR=level).  Both against the oracle (bit-exact), then
the un-chunked stage times of a 512-frame batch (HIP events of the library).  Run once per mode:
    python tools/gpu_desc_blur_ab.py            AOS2_DESC_BLUR=level python tools/gpu_desc_blur_ab.py"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
pkg = g.load_package()
O = g.load_oracle()
mode = os.environ.get("AOS2_DESC_BLUR", "keypoint")
bad = 0
for cfg, seeds in (("tum", (1, 2, 3)), ("kitti", (4,)), ("euroc", (5,))):
    c = pkg.synth.CONFIGS[cfg]
    for sd in seeds:
        img = pkg.synth.synth_image(sd, c["w"], c["h"])
        k, d = pkg.Extractor(nfeatures=c["nfeatures"])(img)
        ok, od = O.Extractor(nfeatures=c["nfeatures"]).extract(img)
        same = len(k) == len(ok) and k.tobytes() == ok.tobytes() and (d == od).all()
        bad += not same
        print(f"[{mode}] {cfg} seed {sd}: {len(k)} keypoints, equal to the oracle: {same}" + ("" if same else f" ({int((d != od).any(1).sum()) if len(k) == len(ok) else -1} descriptors differ)"))
for wh in ((400, 300), (515, 389)):
    img = pkg.synth.synth_image(9, *wh)
    k, d = pkg.Extractor(nfeatures=300)(img)
    ok, od = O.Extractor(nfeatures=300).extract(img)
    same = len(k) == len(ok) and k.tobytes() == ok.tobytes() and (d == od).all()
    bad += not same
    print(f"[{mode}] {wh}: {len(k)} keypoints, equal: {same}")
B, W, H = 512, 640, 480
imgs = np.stack([pkg.synth.synth_image(100 + i, W, H) for i in range(32)])
d_img = torch.from_numpy(imgs[np.arange(B) % 32]).cuda()
ex = pkg.Extractor(nfeatures=1000)
cap = ex.max_keypoints_for(W, H)
kps = torch.empty((B, cap, 7), dtype=torch.float32, device="cuda")
desc = torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda")
n = torch.empty((B,), dtype=torch.int32, device="cuda")
ex.set_chunks(1)
ts = []
for _ in range(8):
    ex.extract_batch_device(d_img.data_ptr(), B, W, H, W, W * H, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr())
    ts.append(ex.last_timing())
med = {k_: float(np.median([t[k_] for t in ts[2:]])) for k_ in ("pyramid", "fast", "octree", "describe", "total_wall")}
print(f"[{mode}] B = 512 un-chunked stage ms (median of 6): {med}")
sys.exit(1 if bad else 0)

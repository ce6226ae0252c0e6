"""Generates the committed golden fixtures from the ORACLE (the reference has no golden vectors of
its own and cannot be built here: SURVEY.md F4/F5, so these pin the oracle against regressions and
give the GPU tests oracle-free expectations).  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
O = g.load_oracle()
S = pkg.synth
OUT = os.path.dirname(os.path.abspath(__file__))


def extractor_case(name, img, nfeatures, nlevels):
    ex = O.Extractor(nfeatures=nfeatures, nlevels=nlevels)
    kps, desc = ex.extract(img)
    stage = {}
    for l in range(nlevels):
        x, y, s = ex.level_candidates(l)
        stage[f"cand_x{l}"], stage[f"cand_y{l}"], stage[f"cand_s{l}"] = x, y, s
        stage[f"pyr_crc{l}"] = np.uint32(zlib.crc32(ex.level_plane(l).tobytes()))
        stage[f"blur_crc{l}"] = np.uint32(zlib.crc32(ex.level_blurred(l).tobytes()))
        stage[f"nkeys{l}"] = np.int32(ex.level_nkeys(l))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), img=img, nfeatures=nfeatures, nlevels=nlevels,
                        kps=kps, desc=desc, **stage)
    print(name, img.shape, len(kps))


extractor_case("extract_320x240_L8", S.synth_image(7, 320, 240), 500, 8)
extractor_case("extract_160x120_L4", S.synth_image(8, 640, 480)[200:320, 300:460].copy(), 200, 4)

# 640x480 / 1241x376: outputs + CRC of the (regenerated) input
for name, (w, h, nf, seed) in dict(tum=(640, 480, 1000, 11), kitti=(1241, 376, 2000, 12)).items():
    img = S.synth_image(seed, w, h)
    kps, desc = O.Extractor(nfeatures=nf).extract(img)
    np.savez_compressed(os.path.join(OUT, f"extract_{name}.npz"), seed=seed, w=w, h=h, nfeatures=nf,
                        img_crc=np.uint32(zlib.crc32(img.tobytes())), kps=kps, desc=desc)
    print(name, len(kps))

# matcher
p = S.synth_bow_problem(21, 300, 320, n_nodes=20, nnratio=0.7)
n, m = O.search_by_bow(p)
np.savez_compressed(os.path.join(OUT, "bow_300.npz"), nmatches=n, match=m, **p)
f, mp = S.synth_proj_mp_problem(22, n_f=300, n_mp=400)
n, m = O.search_by_projection_mp(f, mp)
np.savez_compressed(os.path.join(OUT, "proj_mp_300.npz"), nmatches=n, match=m, **{"f_" + k: v for k, v in f.items()},
                    **{"mp_" + k: v for k, v in mp.items()})
cur, pl = S.synth_proj_last_problem(23, n=300)
n, m = O.search_by_projection_last(cur, pl)
np.savez_compressed(os.path.join(OUT, "proj_last_300.npz"), nmatches=n, match=m, **{"f_" + k: v for k, v in cur.items()},
                    **{"pl_" + k: v for k, v in pl.items()})
# local BA
prob = S.synth_lba_problem(31, n_local=3, n_fixed=2, n_points=50, stereo_frac=0.5)
r = O.lba_solve(prob)
np.savez_compressed(os.path.join(OUT, "lba_3kf.npz"), out_pose_Tcw=r["pose_Tcw"], out_point_xyz=r["point_xyz"],
                    out_outlier=r["edge_outlier"], out_lambda=r["lambda_trace"], out_chi2=r["chi2_trace"], **prob)
prob = S.synth_lba_problem(32, n_local=8, n_fixed=6, n_points=600)
r = O.lba_solve(prob)
np.savez_compressed(os.path.join(OUT, "lba_14kf.npz"), out_pose_Tcw=r["pose_Tcw"], out_point_xyz=r["point_xyz"],
                    out_outlier=r["edge_outlier"], out_lambda=r["lambda_trace"], out_chi2=r["chi2_trace"], **prob)
print("done")
# pose optimisation
prob = S.synth_pose_problem(41, n=500)
r = O.pose_optimization(prob)
np.savez_compressed(os.path.join(OUT, "pose_500.npz"), out_Tcw=r["Tcw"], out_outlier=r["outlier"],
                    out_n_inliers=r["n_inliers"], out_n_bad=r["n_bad"], **prob)
print("pose", r["n_inliers"])
# Frame::ComputeStereoMatches: inputs are regenerated from the seed
seed, w, h, nf = 61, 752, 480, 1200
left, right, disp = S.synth_stereo_pair(seed, w, h)
eL, eR = O.Extractor(nfeatures=nf), O.Extractor(nfeatures=nf)
kl, dl = eL.extract(left)
kr, dr = eR.extract(right)
mbf = np.float32(S.CONFIGS["euroc"]["bf"])
mb = np.float32(mbf / np.float32(S.CONFIGS["euroc"]["fx"]))
ur, dp, n = O.compute_stereo_matches(eL, eR, kl, dl, kr, dr, mb, mbf)
np.savez_compressed(os.path.join(OUT, "stereo_euroc.npz"), seed=seed, w=w, h=h, nfeatures=nf, mb=mb, mbf=mbf,
                    left_crc=np.uint32(zlib.crc32(left.tobytes())), right_crc=np.uint32(zlib.crc32(right.tobytes())),
                    n_left=len(kl), n_right=len(kr), u_right=ur, depth=dp, n_before_cull=n)
print("stereo", n, int((ur >= 0).sum()))
# ORBVocabulary::transform: vocabulary regenerated from the seed
seed, k, L, lu = 51, 10, 3, 2
voc = S.synth_vocabulary(seed, k, L)
V = O.Vocabulary()
V.set_nodes(k, L, 0, 0, voc["parent"], voc["desc"], voc["weight"], voc["is_leaf"])
d = S.vocab_descriptors(np.random.default_rng(52), voc, 600)
np.savez_compressed(os.path.join(OUT, "vocab_k10_L3.npz"), seed=seed, k=k, L=L, levelsup=lu, desc=d, **V.transform(d, lu))
# remaining matcher methods
p = S.synth_bow_kf_problem(24, 300, 320, n_nodes=20, nnratio=0.75)
n, m = O.search_by_bow_kf(p)
np.savez_compressed(os.path.join(OUT, "bow_kf_300.npz"), nmatches=n, match=m, **p)
p = S.synth_triang_problem(25, 300, 320, n_nodes=20)
n, m = O.search_for_triangulation(p)
np.savez_compressed(os.path.join(OUT, "triang_300.npz"), nmatches=n, match=m, **p)
off, desc = S.synth_observations(26, 200, 16)
np.savez_compressed(os.path.join(OUT, "distinctive_200.npz"), off=off, desc=desc, best=O.compute_distinctive_descriptors(off, desc))
f, p = S.synth_proj_gen_problem(41, n_f=400, n_pts=500)
n, bi, bd = O.fuse(f, p)
np.savez_compressed(os.path.join(OUT, "fuse_400.npz"), n=n, best_idx=bi, best_dist=bd, **{"f_" + k: v for k, v in f.items()},
                    **{"p_" + k: v for k, v in p.items()})
n, m = O.search_by_projection_reloc(f, p, 100, True)
np.savez_compressed(os.path.join(OUT, "reloc_400.npz"), n=n, match=m, **{"f_" + k: v for k, v in f.items()},
                    **{"p_" + k: v for k, v in p.items()})
f2, q = S.synth_init_problem(42, 500, 520)
n, m = O.search_for_initialization(f2, q, 100, 0.9, True)
np.savez_compressed(os.path.join(OUT, "init_500.npz"), n=n, match=m, **{"f_" + k: v for k, v in f2.items()},
                    **{"q_" + k: v for k, v in q.items()})

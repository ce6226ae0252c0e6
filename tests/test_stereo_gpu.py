"""Parity of the HIP Frame::ComputeStereoMatches path (src/Frame.cc:495-669) against the oracle and
the golden fixture: mvuRight / mvDepth are compared bit for bit."""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _params(S, cfg):
    mbf = np.float32(S.CONFIGS[cfg]["bf"])
    return np.float32(mbf / np.float32(S.CONFIGS[cfg]["fx"])), mbf


def _oracle_pair(oracle, left, right, nf):
    eL, eR = oracle.Extractor(nfeatures=nf), oracle.Extractor(nfeatures=nf)
    kl, dl = eL.extract(left)
    kr, dr = eR.extract(right)
    return eL, eR, kl, dl, kr, dr


def test_stereo_golden(pkg, gpu):
    g = np.load(os.path.join(GOLD, "stereo_euroc.npz"))
    S = pkg.synth
    left, right, _ = S.synth_stereo_pair(int(g["seed"]), int(g["w"]), int(g["h"]))
    assert np.uint32(zlib.crc32(left.tobytes())) == g["left_crc"]
    xl, xr = pkg.Extractor(nfeatures=int(g["nfeatures"])), pkg.Extractor(nfeatures=int(g["nfeatures"]))
    kl, dl = xl(left)
    kr, dr = xr(right)
    assert len(kl) == int(g["n_left"]) and len(kr) == int(g["n_right"])
    ur, dp = pkg.ComputeStereoMatches(xl, xr, kl, dl, kr, dr, float(g["mb"]), float(g["mbf"]))
    assert ur.tobytes() == g["u_right"].tobytes() and dp.tobytes() == g["depth"].tobytes()


@pytest.mark.parametrize("cfg,seed", [("tum", 1), ("euroc", 2), ("kitti", 3), ("kitti", 4)])
def test_stereo_vs_oracle(pkg, oracle, gpu, cfg, seed):
    S = pkg.synth
    c = S.CONFIGS[cfg]
    mb, mbf = _params(S, cfg)
    left, right, disp = S.synth_stereo_pair(seed, c["w"], c["h"], max_disp=48 if cfg != "kitti" else 96)
    eL, eR, kl, dl, kr, dr = _oracle_pair(oracle, left, right, c["nfeatures"])
    our, odp, on = oracle.compute_stereo_matches(eL, eR, kl, dl, kr, dr, mb, mbf)
    xl, xr = pkg.Extractor(nfeatures=c["nfeatures"]), pkg.Extractor(nfeatures=c["nfeatures"])
    gkl, gdl = xl(left)
    gkr, gdr = xr(right)
    assert gkl.tobytes() == kl.tobytes() and gdr.tobytes() == dr.tobytes()
    ur, dp = pkg.ComputeStereoMatches(xl, xr, gkl, gdl, gkr, gdr, mb, mbf)
    assert (our >= 0).sum() > 200
    assert ur.tobytes() == our.tobytes() and dp.tobytes() == odp.tobytes()
    # size-independent properties (:636-648): depth = mbf / (uL - uR), 0 < disparity < mbf / mb
    m = ur >= 0
    d = gkl["x"][m] - ur[m]
    assert ((dp > 0) == m).all() and (d > 0).all() and (d < mbf / mb).all()
    assert np.allclose(dp[m], mbf / d, rtol=1e-5)


def test_stereo_edge_cases(pkg, oracle, gpu):
    S = pkg.synth
    mb, mbf = _params(S, "tum")
    left, right, _ = S.synth_stereo_pair(5, 320, 240)
    eL, eR, kl, dl, kr, dr = _oracle_pair(oracle, left, right, 500)
    xl, xr = pkg.Extractor(nfeatures=500), pkg.Extractor(nfeatures=500)
    # before any extract: argument error, not a crash
    with pytest.raises(pkg.AosError):
        pkg.ComputeStereoMatches(xl, xr, kl, dl, kr, dr, mb, mbf)
    xl(left)
    xr(right)
    ur, dp = pkg.ComputeStereoMatches(xl, xr, kl, dl, kr[:0], dr[:0], mb, mbf)
    assert (ur == -1).all() and (dp == -1).all()
    ur, dp = pkg.ComputeStereoMatches(xl, xr, kl[:0], dl[:0], kr, dr, mb, mbf)
    assert len(ur) == 0
    # a single left keypoint / odd counts
    for n in (1, 3, 65):
        our, odp, _ = oracle.compute_stereo_matches(eL, eR, kl[:n], dl[:n], kr, dr, mb, mbf)
        ur, dp = pkg.ComputeStereoMatches(xl, xr, kl[:n], dl[:n], kr, dr, mb, mbf)
        assert ur.tobytes() == our.tobytes() and dp.tobytes() == odp.tobytes()
    # unrelated right image
    other = S.synth_image(99, 320, 240)
    ko, do = eR.extract(other)
    gko, gdo = xr(other)
    our, odp, _ = oracle.compute_stereo_matches(eL, eR, kl, dl, ko, do, mb, mbf)
    ur, dp = pkg.ComputeStereoMatches(xl, xr, kl, dl, gko, gdo, mb, mbf)
    assert ur.tobytes() == our.tobytes() and dp.tobytes() == odp.tobytes()
    # identical images: median SAD 0 -> everything culled (reference behaviour)
    eR.extract(left)
    xr(left)
    our, odp, on = oracle.compute_stereo_matches(eL, eR, kl, dl, kl, dl, mb, mbf)
    ur, dp = pkg.ComputeStereoMatches(xl, xr, kl, dl, kl, dl, mb, mbf)
    assert on > 100 and (ur == -1).all() and ur.tobytes() == our.tobytes()
    # left/right extractors that differ -> argument error
    x3 = pkg.Extractor(nfeatures=500, nlevels=4)
    x3(left)
    with pytest.raises(pkg.AosError):
        pkg.ComputeStereoMatches(xl, x3, kl, dl, kr, dr, mb, mbf)


def test_stereo_batch_device_matches_host_api(pkg, oracle, gpu):
    """[batch][cap] device arrays straight from extract_batch_device -> ComputeStereoMatches, no host hop."""
    import torch
    S = pkg.synth
    mb, mbf = _params(S, "euroc")
    B, w, h, nf = 5, 752, 480, 1200
    pairs = [S.synth_stereo_pair(100 + b, w, h) for b in range(B)]
    dev = torch.device("cuda:0")
    L = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
    R = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
    xl, xr = pkg.Extractor(nfeatures=nf), pkg.Extractor(nfeatures=nf)
    cap = xl.max_keypoints
    out = {}
    for name, x, imgs in (("l", xl, L), ("r", xr, R)):
        kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
        desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        n = torch.zeros(B, dtype=torch.int32, device=dev)
        x.extract_batch_device(imgs.data_ptr(), B, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr())
        out[name] = (kps, desc, n)
    ur = torch.zeros((B, cap), dtype=torch.float32, device=dev)
    dp = torch.zeros((B, cap), dtype=torch.float32, device=dev)
    ms = pkg.capi.compute_stereo_matches_device(xl, xr, B, out["l"][0].data_ptr(), out["l"][1].data_ptr(),
                                                out["l"][2].data_ptr(), out["r"][0].data_ptr(), out["r"][1].data_ptr(),
                                                out["r"][2].data_ptr(), cap, mb, mbf, ur.data_ptr(), dp.data_ptr())
    torch.cuda.synchronize()
    assert ms > 0
    nl = out["l"][2].cpu().numpy()
    for b in range(B):
        eL, eR, kl, dl, kr, dr = _oracle_pair(oracle, pairs[b][0], pairs[b][1], nf)
        our, odp, _ = oracle.compute_stereo_matches(eL, eR, kl, dl, kr, dr, mb, mbf)
        assert nl[b] == len(kl)
        assert ur[b, : nl[b]].cpu().numpy().tobytes() == our.tobytes()
        assert dp[b, : nl[b]].cpu().numpy().tobytes() == odp.tobytes()


def test_front_end_chain_extract_stereo_bow_match(pkg, oracle, gpu):
    """The rows compose: ORBextractor (both eyes) -> ComputeStereoMatches -> ORBVocabulary::transform ->
    SearchByBoW, with every intermediate in the format the next stage takes; the device chain equals the
    oracle chain bit for bit."""
    S = pkg.synth
    c = S.CONFIGS["euroc"]
    mb, mbf = _params(S, "euroc")
    left, right, _ = S.synth_stereo_pair(7, c["w"], c["h"])
    nxt = np.roll(left, 9, axis=1)                       # the "next frame": the left image panned by 9 px
    voc = S.synth_vocabulary(3, 10, 4)
    rng = np.random.default_rng(3)
    # device chain
    xl, xr, xn = (pkg.Extractor(nfeatures=c["nfeatures"]) for _ in range(3))
    kl, dl = xl(left)
    kr, dr = xr(right)
    kn, dn = xn(nxt)
    ur, dp = pkg.ComputeStereoMatches(xl, xr, kl, dl, kr, dr, mb, mbf)
    V = pkg.Vocabulary()
    V.set_nodes(voc["k"], voc["L"], 0, 0, voc["parent"], voc["desc"], voc["weight"], voc["is_leaf"])
    fa, fb = V.transform(dl, 4), V.transform(dn, 4)
    has_mp = (dp > 0).astype(np.uint8)                   # map points exist where stereo depth was found

    def pair(fa_, fb_, d1, d2, k1, k2, mp):
        return dict(desc_kf=d1, desc_f=d2, kf_has_mp=mp, angle_kf=k1["angle"].copy(), angle_f=k2["angle"].copy(),
                    node_id_kf=fa_["fv_node"], node_off_kf=fa_["fv_off"], node_idx_kf=fa_["fv_idx"],
                    node_id_f=fb_["fv_node"], node_off_f=fb_["fv_off"], node_idx_f=fb_["fv_idx"],
                    nnratio=np.float32(0.7), check_orientation=1)
    n, match = pkg.Matcher(0.7, True).SearchByBoW(pair(fa, fb, dl, dn, kl, kn, has_mp))
    # oracle chain
    oL, oR, oN = (oracle.Extractor(nfeatures=c["nfeatures"]) for _ in range(3))
    okl, odl = oL.extract(left)
    okr, odr = oR.extract(right)
    okn, odn = oN.extract(nxt)
    our, odp, _ = oracle.compute_stereo_matches(oL, oR, okl, odl, okr, odr, mb, mbf)
    OV = oracle.Vocabulary()
    OV.set_nodes(voc["k"], voc["L"], 0, 0, voc["parent"], voc["desc"], voc["weight"], voc["is_leaf"])
    ofa, ofb = OV.transform(odl, 4), OV.transform(odn, 4)
    on, omatch = oracle.search_by_bow(pair(ofa, ofb, odl, odn, okl, okn, (odp > 0).astype(np.uint8)))
    assert dp.tobytes() == odp.tobytes() and fa["bow_value"].tobytes() == ofa["bow_value"].tobytes()
    assert n == on and (match == omatch).all()
    assert has_mp.sum() > 100                            # the chain is not vacuous
    assert V.score(fa, fb) == oracle.vocab_score_l1(ofa, ofb)

"""Parity of the HIP ORBVocabulary::transform path (DBoW2 TemplatedVocabulary.h:1140-1260) against the
oracle and the golden fixture: BowVector words/values (doubles, bit for bit), FeatureVector CSR, and the
per-feature word / node ids."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("bow_word", "bow_value", "fv_node", "fv_off", "fv_idx", "word_of", "node_of")


def _both(pkg, oracle, voc, scoring=None, weighting=None):
    sc = voc["scoring"] if scoring is None else scoring
    wt = voc["weighting"] if weighting is None else weighting
    V = pkg.Vocabulary()
    V.set_nodes(voc["k"], voc["L"], sc, wt, voc["parent"], voc["desc"], voc["weight"], voc["is_leaf"])
    O = oracle.Vocabulary()
    O.set_nodes(voc["k"], voc["L"], sc, wt, voc["parent"], voc["desc"], voc["weight"], voc["is_leaf"])
    return V, O


def _same(a, b):
    for k in KEYS:
        assert a[k].dtype == b[k].dtype and a[k].tobytes() == b[k].tobytes(), k


def test_transform_golden(pkg, gpu):
    g = np.load(os.path.join(GOLD, "vocab_k10_L3.npz"))
    S = pkg.synth
    voc = S.synth_vocabulary(int(g["seed"]), int(g["k"]), int(g["L"]))
    V = pkg.Vocabulary()
    V.set_nodes(voc["k"], voc["L"], 0, 0, voc["parent"], voc["desc"], voc["weight"], voc["is_leaf"])
    r = V.transform(g["desc"], int(g["levelsup"]))
    for k in KEYS:
        assert r[k].tobytes() == g[k].tobytes(), k


@pytest.mark.parametrize("k,L,levelsup,n", [(10, 3, 1, 1000), (10, 4, 2, 2000), (10, 5, 4, 1500), (4, 6, 4, 700), (18, 3, 2, 300),
                                            (10, 3, 4, 500), (10, 6, 4, 2000)])  # the last one has ORBvoc's shape: 1 111 110 nodes
def test_transform_vs_oracle(pkg, oracle, gpu, k, L, levelsup, n):
    S = pkg.synth
    voc = S.synth_vocabulary(100 + k + L, k, L)
    V, O = _both(pkg, oracle, voc)
    rng = np.random.default_rng(k * 10 + L)
    d = S.vocab_descriptors(rng, voc, n)
    a, b = V.transform(d, levelsup), O.transform(d, levelsup)
    _same(a, b)
    # size-independent properties: L1 norm 1, CSR covers exactly the non-stopped features, ascending keys
    assert abs(a["bow_value"].sum() - 1.0) < 1e-12
    assert (np.diff(a["bow_word"].astype(np.int64)) > 0).all() and (np.diff(a["fv_node"]) > 0).all()
    kept = np.sort(a["fv_idx"])
    assert (np.diff(kept) > 0).all() and len(kept) == a["fv_off"][-1]
    for s in range(len(a["fv_node"])):
        seg = a["fv_idx"][a["fv_off"][s]: a["fv_off"][s + 1]]
        assert (np.diff(seg) > 0).all() and (a["node_of"][seg] == a["fv_node"][s]).all()


def test_transform_weighting_scoring_variants(pkg, oracle, gpu):
    S = pkg.synth
    voc = S.synth_vocabulary(7, 10, 3)
    rng = np.random.default_rng(7)
    d = S.vocab_descriptors(rng, voc, 800)
    d[100:140] = d[100]  # one word hit 40 times: w + w + ... in double
    for scoring in (0, 1, 2, 5):
        for weighting in (0, 1, 2, 3):
            V, O = _both(pkg, oracle, voc, scoring, weighting)
            _same(V.transform(d, 2), O.transform(d, 2))


def test_transform_ragged_tree_and_edge_cases(pkg, oracle, gpu):
    S = pkg.synth
    voc = S.synth_vocabulary(9, 10, 4, ragged=True)  # leaves above level L
    V, O = _both(pkg, oracle, voc)
    rng = np.random.default_rng(9)
    d = S.vocab_descriptors(rng, voc, 1200)
    for lu in (0, 1, 2, 3, 4, 7):
        _same(V.transform(d, lu), O.transform(d, lu))
    # n = 0, 1, non-multiples of 16
    for n in (0, 1, 15, 17, 63):
        a, b = V.transform(d[:n], 2), O.transform(d[:n], 2)
        _same(a, b)
    # all features on stopped words -> both maps empty
    w0 = dict(voc)
    w0["weight"] = np.zeros_like(voc["weight"])
    V0, O0 = _both(pkg, oracle, w0)
    a = V0.transform(d[:50], 2)
    assert len(a["bow_word"]) == 0 and len(a["fv_node"]) == 0 and a["fv_off"].tolist() == [0]
    _same(a, O0.transform(d[:50], 2))
    # empty vocabulary: outputs cleared, no error (TemplatedVocabulary.h:1147-1150)
    E = pkg.Vocabulary()
    assert E.empty()
    a = E.transform(d[:10], 4)
    assert len(a["bow_word"]) == 0 and len(a["fv_node"]) == 0
    # identical children: the first one wins (strict '<', :1240)
    tie = S.synth_vocabulary(11, 10, 2)
    tie["desc"][1:10] = tie["desc"][0]
    Vt, Ot = _both(pkg, oracle, tie)
    _same(Vt.transform(d[:200], 1), Ot.transform(d[:200], 1))


def test_file_loaders_and_score(pkg, oracle, gpu, tmp_path):
    S = pkg.synth
    voc = S.synth_vocabulary(21, 10, 3)
    V, O = _both(pkg, oracle, voc)
    rng = np.random.default_rng(21)
    d1, d2 = S.vocab_descriptors(rng, voc, 900), S.vocab_descriptors(rng, voc, 900)
    d2[:300] = d1[:300]
    ref = V.transform(d1, 2)
    # binary round trip through either writer; the eof quirk adds one duplicate node and word
    pb, ob = tmp_path / "v.bin", tmp_path / "o.bin"
    V.saveToBinaryFile(pb)
    assert O.save_binary(ob) and pb.read_bytes() == ob.read_bytes()
    V2, O2 = pkg.Vocabulary(), oracle.Vocabulary()
    assert V2.loadFromBinaryFile(ob) and O2.load_binary(pb)
    assert V2.info() == O2.info() and V2.info()["nodes"] == V.info()["nodes"] + 1 and V2.info()["words"] == V.info()["words"] + 1
    _same(V2.transform(d1, 2), ref)
    # text format (loadFromTextFile), with and without the final newline
    lines = ["10 3 0 0"]
    for i in range(len(voc["parent"])):
        lines.append(f'{voc["parent"][i]} {int(voc["is_leaf"][i])} ' + " ".join(str(int(x)) for x in voc["desc"][i]) +
                     f' {float(voc["weight"][i])!r}')
    for tail, extra in (("\n", 1), ("", 0)):
        pt = tmp_path / f"v{extra}.txt"
        pt.write_text("\n".join(lines) + tail)
        V3, O3 = pkg.Vocabulary(), oracle.Vocabulary()
        assert V3.loadFromTextFile(pt) and O3.load_text(pt)
        assert V3.info() == O3.info() and V3.info()["nodes"] == V.info()["nodes"] + extra
        _same(V3.transform(d1, 2), O3.transform(d1, 2))
    assert not pkg.Vocabulary().loadFromBinaryFile(tmp_path / "missing.bin")
    # L1 score (KeyFrameDatabase / LoopClosing): self-score 1, symmetric, equal to the oracle's
    a, b = V.transform(d1, 4), V.transform(d2, 4)
    assert abs(V.score(a, a) - 1.0) < 1e-12 and V.score(a, b) == V.score(b, a) == oracle.vocab_score_l1(a, b)
    assert 0.05 < V.score(a, b) < 0.95


def test_transform_device_batch_feeds_search_by_bow(pkg, oracle, gpu):
    """extract_batch_device-shaped descriptor arrays -> transform on the device -> the CSR goes straight into
    SearchByBoW; everything equals the oracle chain."""
    import torch
    S = pkg.synth
    voc = S.synth_vocabulary(31, 10, 5)
    V, O = _both(pkg, oracle, voc)
    rng = np.random.default_rng(31)
    B, cap = 6, 1100
    ns = np.array([1000, 1100, 0, 37, 512, 999], np.int32)
    desc = np.zeros((B, cap, 32), np.uint8)
    for b in range(B):
        desc[b, : ns[b]] = S.vocab_descriptors(rng, voc, int(ns[b]))
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    d_desc, d_n = t(desc), t(ns)
    bw = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    bv = torch.zeros((B, cap), dtype=torch.float64, device=dev)
    fn = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    fo = torch.zeros((B, cap + 1), dtype=torch.int32, device=dev)
    fi = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    nb = torch.zeros(B, dtype=torch.int32, device=dev)
    nf = torch.zeros(B, dtype=torch.int32, device=dev)
    ms = V.transform_device(B, d_desc.data_ptr(), d_n.data_ptr(), cap, 4, bw.data_ptr(), bv.data_ptr(), nb.data_ptr(),
                            fn.data_ptr(), fo.data_ptr(), fi.data_ptr(), nf.data_ptr())
    torch.cuda.synchronize()
    assert ms > 0
    for b in range(B):
        r = O.transform(desc[b, : ns[b]], 4)
        kb, kf = int(nb[b]), int(nf[b])
        assert kb == len(r["bow_word"]) and kf == len(r["fv_node"])
        assert bw[b, :kb].cpu().numpy().view(np.uint32).tobytes() == r["bow_word"].tobytes()
        assert bv[b, :kb].cpu().numpy().tobytes() == r["bow_value"].tobytes()
        assert fn[b, :kf].cpu().numpy().tobytes() == r["fv_node"].tobytes()
        assert fo[b, : kf + 1].cpu().numpy().tobytes() == r["fv_off"].tobytes()
        assert fi[b, : r["fv_off"][-1]].cpu().numpy().tobytes() == r["fv_idx"].tobytes()

"""Writes the cases tools/refcheck's binaries read (bundles in the layout of tests/bundle_io.py), each holding a seeded problem AND
the oracle's result for it:
    python tools/refcheck/dump_cases.py /tmp/refcases
    lba_<i>.bundle      LocalBundleAdjustment problems (synth.synth_lba_problem / lba_window_mix) + oracle poses / points / outlier flags
    bow_<i>.bundle      SearchByBoW pairs (two images, their extraction is the binary's job) + the vocabulary file voc.txt + the oracle's matches
    dist.bundle         descriptor pairs + the oracle's DescriptorDistance
Runs in this container (no GPU, no OpenCV): the oracle is the C restatement."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g   # noqa: E402
import bundle_io               # noqa: E402


def main(out):
    os.makedirs(out, exist_ok=True)
    pkg, O = g.load_package(), g.load_oracle()
    import parity   # oracle/parity.py
    S = pkg.synth
    cases = [dict(seed=1, n_local=3, n_fixed=2, n_points=60, stereo_frac=0.5), dict(seed=2, n_local=6, n_fixed=4, n_points=400, stereo_frac=0.0),
             dict(seed=4, n_local=5, n_fixed=0, n_points=300, include_kf0=True), dict(seed=0), dict(seed=3, include_kf0=True, outlier_frac=0.15)]
    cases += S.lba_window_mix(0, 4)
    cases += [m for m in S.lba_window_mix(0, 16, hard_every=8) if "hard" in m]   # two windows that start far off the optimum (rejected steps)
    for i, kw in enumerate(cases):
        p = S._lba_from_kwargs(kw)
        w = O.lba_solve(p)
        arrs = {k: p[k] for k in ("pose_Tcw", "pose_fixed", "pose_id", "point_xyz", "point_id", "edge_pose", "edge_point", "edge_obs", "edge_stereo", "edge_inv_sigma2")}
        arrs["pose_id"] = np.asarray(p["pose_id"], np.int64)
        arrs["point_id"] = np.asarray(p["point_id"], np.int64)
        arrs["cam"] = np.array([p["fx"], p["fy"], p["cx"], p["cy"], p["bf"]], np.float32)
        arrs["out_pose_Tcw"], arrs["out_point_xyz"], arrs["out_outlier"] = w["pose_Tcw"], w["point_xyz"], w["edge_outlier"].astype(np.uint8)
        arrs["out_iters"] = np.asarray(w["iters"], np.int32)
        # the oracle's own resolution on this window (parity.lba_resolution: its re-associated runs against itself) -- a pinned run's
        # difference from out_* is a disagreement only beyond this
        res = parity.lba_resolution(p, want=w)
        arrs["resolution"] = np.array([res["pose"], res["point"], 1.0 if res["decisions_equal"] else 0.0], np.float64)
        bundle_io.save(os.path.join(out, f"lba_{i}.bundle"), arrs)
        print(f"lba_{i}: {p['n_poses']} keyframes, {p['n_points']} points, {p['n_edges']} edges, oracle iterations {w['iters']}, "
              f"oracle against its re-associated runs: poses {res['pose']:.2e} points {res['point']:.2e}")
    rng = np.random.default_rng(5)
    a, b = S.synth_descriptors(rng, 4096), S.synth_descriptors(rng, 4096)
    bundle_io.save(os.path.join(out, "dist.bundle"), dict(a=a, b=b, dist=np.array([O.descriptor_distance(x, y) for x, y in zip(a, b)], np.int32)))
    # SearchByBoW: two views of a scene, a small vocabulary in DBoW2's text format (k L scoring weighting / parent is_leaf 32 bytes weight)
    voc = S.synth_vocabulary(400, 10, 3)
    with open(os.path.join(out, "voc.txt"), "w") as f:
        f.write(f"{voc['k']} {voc['L']} {voc['scoring']} {voc['weighting']}\n")
        for pa, d, wt, lf in zip(voc["parent"], voc["desc"], voc["weight"], voc["is_leaf"]):
            f.write(f"{int(pa)} {int(lf)} " + " ".join(str(int(v)) for v in d) + f" {float(wt)!r}\n")
    ov = O.Vocabulary()
    ov.set_nodes(voc["k"], voc["L"], voc["scoring"], voc["weighting"], voc["parent"], voc["desc"], voc["weight"], voc["is_leaf"])
    for i in range(3):
        sc = pkg.scenario.tracking_scenario(40 + i, 1, n_unique=1)
        oe = O.Extractor(nfeatures=1000)
        k1, d1 = oe.extract(sc["last"][0])
        k2, d2 = oe.extract(sc["cur"][0])
        b1, b2 = ov.transform(d1, 4), ov.transform(d2, 4)
        has_mp = (np.random.default_rng(i).random(len(k1)) < 0.8).astype(np.uint8)
        prob = dict(desc_kf=d1, desc_f=d2, kf_has_mp=has_mp, angle_kf=k1["angle"], angle_f=k2["angle"], node_id_kf=b1["fv_node"], node_off_kf=b1["fv_off"],
                    node_idx_kf=b1["fv_idx"], node_id_f=b2["fv_node"], node_off_f=b2["fv_off"], node_idx_f=b2["fv_idx"], nnratio=np.float32(0.7), check_orientation=1)
        n, m = O.search_by_bow(prob)
        bundle_io.save(os.path.join(out, f"bow_{i}.bundle"), dict(img_kf=sc["last"][0], img_f=sc["cur"][0], depth=sc["depth_last"][0], kf_has_mp=has_mp,
                                                                 cam=np.array([sc["fx"], sc["fy"], sc["cx"], sc["cy"], sc["mbf"]], np.float32),
                                                                 out_n=np.array([n], np.int32), out_match=np.asarray(m, np.int32),
                                                                 out_kp_kf=np.stack([k1["x"], k1["y"]], 1).astype(np.float32), out_kp_f=np.stack([k2["x"], k2["y"]], 1).astype(np.float32)))
        print(f"bow_{i}: {len(k1)} / {len(k2)} features, oracle SearchByBoW {n} matches")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/refcases")

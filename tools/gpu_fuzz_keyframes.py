"""One-off parity sweep (a time-boxed slice of it runs in tests/test_fuzz_gpu.py) of the device-resident keyframe work -- SearchForTriangulation
and the search part of Fuse over (keyframe, neighbour) pairs of device-resident keyframe batches (chain.KeyFrameWork) -- vs the oracle
on its own extraction of the same images.  python tools/gpu_fuzz_keyframes.py [n_cases [seconds]]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
sys.path.insert(0, os.path.dirname(O.__file__))
import parity
budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 1e18
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(97531)
bad, ran, pairs = 0, 0, 0
t0 = time.time()
for c in range(n_cases):
    if time.time() - t0 > budget_s:
        break
    ran += 1
    nu = int(rng.integers(1, 4)); B = int(rng.integers(nu, 7)); n_kf = int(rng.integers(1, B + 1)); n_nb = int(rng.choice([1, 2, 5]))
    k, L = int(rng.choice([4, 10])), int(rng.choice([3, 4, 5]))
    levelsup = int(rng.integers(1, L + 1))
    params = dict(seed=7000 + c, B=B, nu=nu, n_kf=n_kf, n_nb=n_nb, k=k, L=L, levelsup=levelsup, only_stereo=bool(rng.integers(0, 2)),
                  ori=bool(rng.integers(0, 2)), th=float(rng.choice([2.5, 3.0, 4.0])), max_shift=int(rng.choice([4, 10, 16])))
    scen = pkg.scenario.tracking_scenario(params["seed"], B, n_unique=nu, max_shift=params["max_shift"])
    tc = pkg.chain.TrackingChain(scen, n_local=600)
    voc = pkg.synth.synth_vocabulary(9000 + c, k, L)
    P = n_kf * n_nb
    # a finite epipole inside the image for half of the cases: drives the mono-mono proximity test (src/ORBmatcher.cc:739-745)
    ep = None if c % 2 else np.stack([rng.uniform(0, scen["w"], P), rng.uniform(0, scen["h"], P)], 1).astype(np.float32)
    kw = pkg.chain.KeyFrameWork(tc, voc, n_kf=n_kf, n_nb=n_nb, levelsup=levelsup, fuse_th=params["th"], only_stereo=params["only_stereo"],
                                check_orientation=params["ori"], epipole=ep, nb_cap=tc.cap + int(rng.choice([0, 8, 72])))
    co = parity.ChainOracle(scen, tc)
    kw.run()
    m = parity.keyframe_work_mismatches(kw, co, voc, range(P))
    pairs += P
    if m:
        bad += len(m)
        print("MISMATCH case", c, params, m[:3], flush=True)
    for x in (kw, tc):
        del x
print("cases", ran, "of", n_cases, "pairs", pairs, "MISMATCHES", bad, "time %.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)

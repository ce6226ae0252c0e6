import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
sys.path.insert(0, os.path.dirname(O.__file__))
import parity
nk = int(os.environ.get("NK", "24"))
scen = pkg.scenario.tracking_scenario(31, nk, n_unique=nk)
tc = pkg.chain.TrackingChain(scen, n_local=1500)
voc = pkg.synth.synth_vocabulary(400, 10, int(os.environ.get("VL", "4")))
kw2 = pkg.chain.KeyFrameWork(tc, voc, n_kf=nk, n_nb=10, n_second=2)
co = parity.ChainOracle(scen, tc)
kw2.run()
q = kw2.snapshot()
# the neighbours' extraction on the device against the oracle's
n_host = kw2.n_n.cpu().numpy()
oe = O.Extractor(nfeatures=scen["nfeatures"])
badj = [j for j in range(0, len(n_host), 7) if len(oe.extract(kw2.nb["imgs"][j])[0]) != n_host[j]]
print("neighbour frames whose keypoint count differs from the oracle's (every 7th checked):", badj[:10], len(n_host))
bad = parity.keyframe_work_mismatches(q, co, voc, range(len(kw2.kf1)))
print("pairs", len(kw2.kf1), "tri", len(kw2.tri_pairs), "mismatches", len(bad))
for b in bad[:4]: print("  ", b)
import collections
c = collections.Counter(b.split(":")[1].split(" ")[1] for b in bad)
print(c)
firstbad = [b for b in bad if "Triangulation" in b][:3]
print(firstbad)

import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg=g.load_package()
ba=pkg.LocalBA()
p=pkg.synth.synth_lba_problem(0)
ba.LocalBundleAdjustment(p)
r=ba.LocalBundleAdjustment(p)
print(r["iters"], r["ms_device"])

"""One LocalBA batch of N windows, solved 3 times (run under rocprofv3 --kernel-trace --stats: tools/prof_lba_mix.sh).
LBA_MIX = het   : synth.lba_window_mix (bench.py's default step: 10-40 local keyframes, 2-6 k points)
          het26 : the same mix with the local keyframes capped at 26 (every reduced system fits LDS)
          hom   : round 3's batch (SURVEY 8(d)-size windows, 4 distinct, tiled)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
N = int(os.environ.get("LBA_N", "64"))
mode = os.environ.get("LBA_MIX", "het")
if mode == "hom":
    u = [pkg.synth.synth_lba_problem(i, n_points=8000) for i in range(4)]
    probs = [u[i % 4] for i in range(N)]
else:
    mix = pkg.synth.lba_window_mix(0, N)
    if mode == "het26":
        for m in mix:
            m["n_local"] = min(m["n_local"], 26)
    probs = pkg.synth.synth_lba_problems(mix)
ba = pkg.LocalBA()
prep = ba.prepare_batch(probs)
ba.solve_prepared(prep)
for _ in range(3):
    t0 = time.time()
    ba.solve_prepared(prep)
    print("%s: %d windows, wall %.2f ms dev %.2f ms, program %s, edges %d..%d" % (mode, N, (time.time() - t0) * 1e3, prep["R"][0].ms_device, ba.last_program(),
          min(p["n_edges"] for p in probs), max(p["n_edges"] for p in probs)), file=sys.stderr)

# copies the summaries of tools/prof_r06.sh / fuzz_r06.sh (gpurun_out/) into profiles/ under r06_ names
cd "$(dirname "$0")/.."; O=gpurun_out/prof_r06; P=profiles
cpi() { [ -f "$1" ] && cp "$1" "$2"; }
cpi $O/pytest_gpu.log $P/r06_pytest_gpu.log
cpi $O/bench_kernel_stats.csv $P/r06_bench_b512_kernel_stats.csv
cpi $O/bench_fast_kernel_trace.txt $P/r06_bench_fast_kernel_trace.txt
[ -f $O/bench_profiled.json ] && tail -1 $O/bench_profiled.json > $P/r06_bench_profiled_run.json
[ -f $O/bench_kitti.json ] && tail -1 $O/bench_kitti.json > $P/r06_bench_kitti.json
cpi $O/kitti_kernel_stats.csv $P/r06_bench_kitti_kernel_stats.csv
grep -v "^+" $O/pmc_extract_b512.txt | sed 's#/tmp/code/[^ ]*/gpurun_out/#gpurun_out/#' > $P/r06_pmc_extract_b512.txt
cpi $O/extract_b512_kernel_stats.csv $P/r06_extract_b512_kernel_stats.csv
cpi $O/extractor_counters.json $P/r06_extractor_counters.json
grep -v "^+" $O/pmc_lba.txt | sed 's#/tmp/code/[^ ]*/gpurun_out/#gpurun_out/#' > $P/r06_pmc_lba.txt
cpi $O/lba_het64_kernel_stats.csv $P/r06_lba_het64_kernel_stats.csv
cpi $O/lba_counters.json $P/r06_lba_counters.json
[ -f $O/pmc_extract_kitti_b256.txt ] && grep -v "^+" $O/pmc_extract_kitti_b256.txt | sed 's#/tmp/code/[^ ]*/gpurun_out/#gpurun_out/#' > $P/r06_pmc_extract_kitti_b256.txt
cpi $O/extract_kitti_b256_kernel_stats.csv $P/r06_extract_kitti_b256_kernel_stats.csv
cpi $O/extractor_counters_kitti.json $P/r06_extractor_counters_kitti.json
cpi $O/lba_unprofiled.txt $P/r06_lba_unprofiled.txt
cpi $O/lba_determinism.txt $P/r06_lba_determinism.txt
grep -v amdgpu.ids $O/chain_latency.txt > $P/r06_chain_latency.txt
for f in host_images homogeneous python_threads; do [ -f $O/bench_$f.json ] && tail -1 $O/bench_$f.json > $P/r06_bench_n1_$f.json; done
# the default line: the five consecutive runs' values, the median run as the committed line
python - <<'PY'
import json, glob
runs = []
for f in sorted(glob.glob("gpurun_out/prof_r06/bench_default_*.json")):
    try:
        runs.append((json.loads(open(f).read().strip().splitlines()[-1]), f))
    except Exception as e:
        print("skip", f, e)
if runs:
    runs_sorted = sorted(runs, key=lambda r: r[0]["value"])
    med = runs_sorted[len(runs_sorted) // 2]
    json.dump(med[0], open("profiles/r06_bench_n1.json", "w"))
    with open("profiles/r06_bench_n1_spread.txt", "w") as o:
        o.write("python bench.py, consecutive runs right behind the gpu suite on one box (tools/prof_r06.sh): frames/s, ms per step, parity_checked.ok\n")
        for d, f in runs:
            o.write("%s  %.0f  %.3f  %s\n" % (f.split("/")[-1], d["value"], d["ms_per_step"], d["parity_checked"]["ok"]))
        v = [d["value"] for d, _ in runs]
        o.write("min %.0f median %.0f max %.0f: +%.1f %% / -%.1f %% around the median\n" % (min(v), med[0]["value"], max(v), 100 * (max(v) / med[0]["value"] - 1), 100 * (1 - min(v) / med[0]["value"])))
    print(open("profiles/r06_bench_n1_spread.txt").read())
PY
mkdir -p $P/r06_fuzz
for f in gpurun_out/fuzz_r06/*.txt; do [ -f "$f" ] && cp "$f" $P/r06_fuzz/; done
git rev-parse HEAD > $P/r06_fuzz/COMMIT.txt
ls $P | grep r06

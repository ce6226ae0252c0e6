cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_chain_now
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/tools/gpu_chain_batch_prof.py > $O/log.txt 2>&1
tail -1 $O/log.txt
f=$(find $O -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:18]:
    print(r['Name'][:60].ljust(60), r['Calls'], "%.1f us avg" % (float(r['AverageNs'])/1e3), r['Percentage'])
PY
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete

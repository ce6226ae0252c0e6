cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_bench_now
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra > $O/log.txt 2>&1
tail -1 $O/log.txt | cut -c1-200
f=$(find $O -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("sum of kernel durations %.1f ms over 22 steps (+setup) = %.2f ms per step"%(tot/1e6, tot/1e6/22))
for r in rows[:16]:
    print(r['Name'][:58].ljust(58), r['Calls'].rjust(5), "%7.1f us avg"%(float(r['AverageNs'])/1e3), "%6.2f ms/step"%(float(r['TotalDurationNs'])/1e6/22), r['Percentage'])
PY
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete

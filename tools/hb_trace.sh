# memory-copy + kernel trace of bench.py --host-images (which copies run when, how long, beside which kernels)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/hb
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/hb -- python $R/bench.py --host-images --no-extra --no-cpu-baseline --steps 12 --warmup 4 > $R/gpurun_out/hb_line.json 2> $R/gpurun_out/hb_err.txt
cd $R
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/hb/*/*memory_copy_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
big = [r for r in rows if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 200000]
t0 = int(rows[0]['Start_Timestamp'])
for r in big[-40:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%10.2f ms  dur %8.2f ms  %s %s" % ((s - t0) / 1e6, (e - s) / 1e6, r.get('Direction', ''), r.get('Bytes', r.get('Size', ''))))
import collections
c = collections.Counter()
for r in rows: c[r.get('Direction','')] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print({k: v / 1e6 for k, v in c.items()}, len(rows))
PY
find gpurun_out/hb -name "*kernel_trace.csv" -size +30M -delete

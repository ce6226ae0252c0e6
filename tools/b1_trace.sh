# kernel trace of the single-sequence chain (B = 1): per-kernel average durations and the timeline of one full step
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/b1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b1 -- python $R/tools/gpu_chain_latency.py > $R/gpurun_out/b1_out.txt 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/b1/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last full step: find the last occurrence of the first kernel of a step (resize of level 1) going backwards ~60 kernels
names = [r['Kernel_Name'] for r in rows]
# print the last 70 kernels with start offsets
last = rows[-75:]
t0 = int(last[0]['Start_Timestamp'])
prev_end = t0
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%9.1f us  dur %7.1f  gap %6.1f  %s  grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r['Kernel_Name'][:60], r.get('Grid_Size', '')))
    prev_end = max(prev_end, e)
PY

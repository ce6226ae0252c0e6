cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c7
for e in 0 1 2 3 5; do echo "EXTRA_STREAMS=$e"; EXTRA_STREAMS=$e python tools/gpu_pipelined_trace.py 2>&1 | grep "per frame"; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c7/tr -- python $GRAFT_REPO_ROOT/tools/gpu_pipelined_trace.py 2>&1 | grep "per frame"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/c7/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
qs = sorted(set(r['Queue_Id'] for r in rows))
rows = rows[-70:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('aos2::', '').replace('void ', '')
    print('q%d %-34s %8.1f -> %8.1f (%6.1f us)' % (qs.index(r['Queue_Id']), n[:34], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
find gpurun_out/c7 -name "*.csv" -delete; find gpurun_out/c7 -name "*.db" -delete

cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/b1g
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/b1g -- python $R/tools/gpu_graph_chain.py > $R/gpurun_out/b1g_out.txt 2>&1
cd $R; tail -4 gpurun_out/b1g_out.txt
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/b1g/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
starts=[i for i,n in enumerate(names) if 'pyramid_fused' in n]
a,b=starts[-3],starts[-2]
t0=int(rows[a]['Start_Timestamp']); prev=t0; tot=0
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print("%8.1f  dur %6.1f gap %6.1f  %s" % ((s-t0)/1e3,(e-s)/1e3,(s-prev)/1e3,r['Kernel_Name'][:58]))
    prev=max(prev,e); tot+=(e-s)
print("span %.1f us, busy %.1f us, kernels %d, period %.1f" % ((prev-t0)/1e3, tot/1e3, b-a, (int(rows[b]['Start_Timestamp'])-t0)/1e3))
PY

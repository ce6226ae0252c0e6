cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c20
for i in 2 3; do python bench.py > gpurun_out/c20/bench_$i.json 2> gpurun_out/c20/err_$i; python -c "
import json; d=json.loads(open('gpurun_out/c20/bench_$i.json').read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), d['parity_checked']['ok'], d['extra']['timed_steps']['host_thread_waits_ms_per_step'])"; done

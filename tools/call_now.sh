# scratch: the command of the last gpurun call of the round (gpu suite + smoke + three default bench lines at the final commit)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final/pytest_gpu.log 2>&1; tail -2 gpurun_out/final/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2 3; do python bench.py > gpurun_out/final/bench_$i.json 2> gpurun_out/final/err_$i; python -c "
import json; d=json.loads(open('gpurun_out/final/bench_$i.json').read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), d['parity_checked']['ok'], d['extra']['timed_steps']['host_thread_waits_ms_per_step'])"; done

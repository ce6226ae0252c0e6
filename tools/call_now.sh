cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do
python bench.py --no-cpu-baseline 2>gpurun_out/c5/bench.err | tail -1 > gpurun_out/c5/bench_$i.json; python -c "
import json; d=json.loads(open('gpurun_out/c5/bench_$i.json').read()); print(d['value'], d['ms_per_step'], d['parity_checked']['ok'], d['parity_checked']['n_mismatches'], d['extra'].get('composite_with_homogeneous_local_ba_windows',{}).get('frames_per_s')); print(d['extra'].get('single_sequence'))"
done
python tools/gpu_chain_latency.py 2>&1 | grep -v amdgpu.ids | head -3

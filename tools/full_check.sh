# gpu suite (or a subset: TESTS=...) + the default bench line
cd $GRAFT_REPO_ROOT
python -m pytest ${TESTS:-tests} -m gpu -x -q 2>&1 | tail -4
python bench.py ${BENCH_ARGS} 2>/dev/null | tail -1 > gpurun_out/bench_now.json; python -c "
import json; d=json.loads(open('gpurun_out/bench_now.json').read()); print(d['value'], d['ms_per_step'], d['parity_checked']['ok'], d['parity_checked']['n_mismatches'], d['stage_ms'], d['extra'].get('composite_with_homogeneous_local_ba_windows',{}).get('frames_per_s'))"

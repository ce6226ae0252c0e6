cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c21
timeout 900 python -m pytest tests/test_frames_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_bench_gpu.py -m gpu -x -q -k "contract or two_ranks" 2>&1 | tail -3
for i in 1 2 3 4 5 6; do
  python bench.py --no-extra --no-cpu-baseline --no-verify 2> gpurun_out/c21/err | tail -1 > gpurun_out/c21/r_$i.json
  python -c "
import json; d=json.loads(open('gpurun_out/c21/r_$i.json').read()); t=d['extra']['timed_steps']; print('value %.0f ms_per_step %.3f' % (d['value'], d['ms_per_step']), t['host_thread_waits_ms_per_step'], 'kf', t['keyframe_job_wall_ms_min_median_max'])" || tail -3 gpurun_out/c21/err
done

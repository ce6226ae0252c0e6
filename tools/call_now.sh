cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c13
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3 4 5 6 7 8; do
  python bench.py --no-extra --no-cpu-baseline --no-verify 2> gpurun_out/c13/err | tail -1 > gpurun_out/c13/r_$i.json
  python -c "
import json; d=json.loads(open('gpurun_out/c13/r_$i.json').read()); t=d['extra']['timed_steps']; print('value %.0f ms_per_step %.3f' % (d['value'], d['ms_per_step']), t['host_thread_waits_ms_per_step'], 'step', t['step_to_step_ms_min_median_max'], 'lba', t['local_ba_call_wall_ms_min_median_max'], 'kf', t['keyframe_job_wall_ms_min_median_max'])" || tail -3 gpurun_out/c13/err
done

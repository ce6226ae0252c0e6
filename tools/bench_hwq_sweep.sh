# does the number of hardware queues the runtime maps the streams onto limit the overlap?  (the step uses ~12 streams; the runtime's default is 4 queues)
# first sweep (round 4, one box): 4 queues 50.2 k, 8 queues 60.0 k, 16 queues 61.2 k frames/s with two steps in flight; three steps in flight 55.3 / 55.0 / 54.8 k
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-"4 2 2" "16 2 2" "16 2 2" "32 2 2" "16 2 3" "16 2 4" "4 2 2"}; do
  set -- $cfg
  echo "GPU_MAX_HW_QUEUES=$1 INFLIGHT=$2 LBA_HANDLES=$3"
  GPU_MAX_HW_QUEUES=$1 AOS2_BENCH_INFLIGHT=$2 AOS2_BENCH_LBA_HANDLES=$3 python bench.py --no-extra --no-cpu-baseline --no-verify 2> gpurun_out/hwq.err | tail -1 > gpurun_out/hwq.json
  python -c "
import json; d=json.loads(open('gpurun_out/hwq.json').read()); print('   value %.0f ms_per_step %.3f' % (d['value'], d['ms_per_step']))" || tail -3 gpurun_out/hwq.err
done

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c8
run() {
  echo "Q=$1 $2"
  env GPU_MAX_HW_QUEUES=$1 $2 python bench.py --no-extra --no-cpu-baseline --no-verify 2> gpurun_out/c8/hwq.err | tail -1 > gpurun_out/c8/r.json
  python -c "
import json; d=json.loads(open('gpurun_out/c8/r.json').read()); print('   value %.0f ms_per_step %.3f' % (d['value'], d['ms_per_step']))" || tail -3 gpurun_out/c8/hwq.err
}
run 4 A=1
run 2 A=1
run 3 A=1
run 1 A=1
run 4 AOS2_LBA_STREAM_PRIORITY=normal
run 2 AOS2_LBA_STREAM_PRIORITY=normal

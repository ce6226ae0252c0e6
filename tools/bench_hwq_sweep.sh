# does the number of hardware queues the runtime maps the streams onto limit the overlap?  (the step uses ~12 streams; the default is 4 queues)
cd $GRAFT_REPO_ROOT
for q in 4 8 16; do
  for inflight in 2 3; do
    echo "GPU_MAX_HW_QUEUES=$q INFLIGHT=$inflight"
    GPU_MAX_HW_QUEUES=$q AOS2_BENCH_INFLIGHT=$inflight python bench.py --no-extra --no-cpu-baseline --no-verify 2> gpurun_out/hwq_$q.err | tail -1 > gpurun_out/hwq_$q.json
    python -c "
import json; d=json.loads(open('gpurun_out/hwq_$q.json').read()); print('   value %.0f ms_per_step %.3f' % (d['value'], d['ms_per_step']))" || tail -3 gpurun_out/hwq_$q.err
  done
done

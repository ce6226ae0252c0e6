cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c12
run() {
  echo "== NUMA=$1"
  env AOS2_BENCH_NUMA=$1 python bench.py --no-extra --no-cpu-baseline --no-verify 2> gpurun_out/c12/err | tail -1 > gpurun_out/c12/r.json
  python -c "
import json; d=json.loads(open('gpurun_out/c12/r.json').read()); print('   value %.0f ms_per_step %.3f bound %s' % (d['value'], d['ms_per_step'], d['config'].get('host_cpus_bound_to_the_gpus_numa_node')))" || tail -3 gpurun_out/c12/err
}
for rep in 1 2 3 4; do for m in setup 0; do run $m; done; done

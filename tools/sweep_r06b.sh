# wave priority of the LocalBA kernels: the composite and the batch alone, per library variant
cd $GRAFT_REPO_ROOT; O=gpurun_out/sweep_r06b; mkdir -p $O; L=$PWD/active-orb-slam2_amd/lib
run() { name=$1; shift; env "$@" python bench.py --no-extra --no-cpu-baseline --no-verify --steps 60 --warmup 6 2>$O/$name.err | tail -1 > $O/$name.json
  python - <<PY
import json
d = json.loads(open("$O/$name.json").read()); t = d["extra"]["timed_steps"]
print("%-28s %8.0f frames/s  %.3f ms/step  waits %s  lba wall %s  kf wall %s" % ("$name", d["value"], d["ms_per_step"], t["host_thread_waits_ms_per_step"], t["local_ba_call_wall_ms_min_median_max"], t["keyframe_job_wall_ms_min_median_max"]))
PY
}
for v in lbaprio0 "" lbaprio3 lbaprio0 ""; do
  lib=$L/libaos2${v:+_$v}.so
  run "composite_${v:-prio2}" AOS2_LIB=$lib
  AOS2_LIB=$lib python tools/gpu_lba_mix_prof.py 2>&1 | tail -1
done

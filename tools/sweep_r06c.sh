cd $GRAFT_REPO_ROOT; O=gpurun_out/sweep_r06c; mkdir -p $O
run() { name=$1; shift; env "$@" python bench.py --no-extra --no-cpu-baseline --no-verify --steps 60 --warmup 6 2>$O/$name.err | tail -1 > $O/$name.json
  python - <<PY
import json
d = json.loads(open("$O/$name.json").read()); t = d["extra"]["timed_steps"]
print("%-28s %8.0f frames/s  %.3f ms/step  waits %s  lba wall %s  kf wall %s" % ("$name", d["value"], d["ms_per_step"], t["host_thread_waits_ms_per_step"], t["local_ba_call_wall_ms_min_median_max"], t["keyframe_job_wall_ms_min_median_max"]))
PY
}
run default X=1
run ldlt_old AOS2_LDLT=old
run ldlt_old_nokf AOS2_LDLT=old AOS2_BENCH_NO_BOW=1
run default_nokf AOS2_BENCH_NO_BOW=1
run hwq8 GPU_MAX_HW_QUEUES=8
run ldlt_old2 AOS2_LDLT=old
run default2 X=1

# composite decomposition: which part of the step is the device busy with (each run ~40 s)
cd $GRAFT_REPO_ROOT; O=gpurun_out/sweep_r06a; mkdir -p $O
run() { name=$1; shift; env "$@" python bench.py --no-extra --no-cpu-baseline --no-verify --steps 60 --warmup 6 2>$O/$name.err | tail -1 > $O/$name.json
  python - <<PY
import json
d = json.loads(open("$O/$name.json").read()); t = d["extra"]["timed_steps"]
print("%-28s %8.0f frames/s  %.3f ms/step  waits %s  lba wall %s  kf wall %s" % ("$name", d["value"], d["ms_per_step"], t["host_thread_waits_ms_per_step"], t["local_ba_call_wall_ms_min_median_max"], t["keyframe_job_wall_ms_min_median_max"]))
PY
}
run default X=1
run no_lba AOS2_BENCH_NO_LBA=1
run no_kfw AOS2_BENCH_NO_KEYFRAME_WORK=1
run no_bow_no_kfw AOS2_BENCH_NO_BOW=1
run no_lba_no_bow AOS2_BENCH_NO_LBA=1 AOS2_BENCH_NO_BOW=1
run prio_kf AOS2_PRIO_MATCHER=1 AOS2_PRIO_VOCABULARY=1
run prio_kf_frames AOS2_PRIO_MATCHER=1 AOS2_PRIO_VOCABULARY=1 AOS2_PRIO_FRAMES=1
run lba_normal_prio AOS2_LBA_STREAM_PRIORITY=normal
run default_again X=1

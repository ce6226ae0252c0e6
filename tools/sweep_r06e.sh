cd $GRAFT_REPO_ROOT; O=gpurun_out/sweep_r06e; mkdir -p $O
python -m pytest tests/test_bench_gpu.py tests/test_frames_gpu.py -m gpu -x -q 2>&1 | tail -5
run() { name=$1; shift; env "$@" python bench.py --no-extra --no-cpu-baseline --steps 60 --warmup 6 2>$O/$name.err | tail -1 > $O/$name.json
  python - <<PY
import json
d = json.loads(open("$O/$name.json").read()); t = d["extra"]["timed_steps"]
print("%-12s %8.0f frames/s  %.3f ms/step  parity %s  waits %s  lba wall %s  kf wall %s" % ("$name", d["value"], d["ms_per_step"], d["parity_checked"]["ok"], t["host_thread_waits_ms_per_step"], t["local_ba_call_wall_ms_min_median_max"], t["keyframe_job_wall_ms_min_median_max"]))
PY
}
run spc2 X=1
run spc1 AOS2_BENCH_LBA_STEPS_PER_CALL=1
run spc3 AOS2_BENCH_LBA_STEPS_PER_CALL=3
run spc4 AOS2_BENCH_LBA_STEPS_PER_CALL=4
run spc2_4h AOS2_BENCH_LBA_HANDLES=4
run spc2b X=1
run spc1b AOS2_BENCH_LBA_STEPS_PER_CALL=1

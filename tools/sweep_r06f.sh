cd $GRAFT_REPO_ROOT; O=gpurun_out/sweep_r06f; mkdir -p $O
for v in 0 1 0 1; do echo "AOS2_OCT_IMAGE=$v:"; AOS2_OCT_IMAGE=$v python tools/prof_extract.py 512 2>&1 | tail -1; done
for v in 0 1; do echo "kitti AOS2_OCT_IMAGE=$v:"; AOS2_OCT_IMAGE=$v python tools/prof_extract.py 256 kitti 2>&1 | tail -1; done
run() { name=$1; shift; env "$@" python bench.py --no-extra --no-cpu-baseline --no-verify --steps 60 --warmup 6 2>$O/$name.err | tail -1 > $O/$name.json
  python - <<PY
import json
d = json.loads(open("$O/$name.json").read()); t = d["extra"]["timed_steps"]
print("%-12s %8.0f frames/s  %.3f ms/step  waits %s" % ("$name", d["value"], d["ms_per_step"], t["host_thread_waits_ms_per_step"]))
PY
}
run default X=1
run octimage AOS2_OCT_IMAGE=1
run default2 X=1
run octimage2 AOS2_OCT_IMAGE=1

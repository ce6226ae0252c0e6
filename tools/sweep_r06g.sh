cd $GRAFT_REPO_ROOT; O=gpurun_out/sweep_r06g; mkdir -p $O
python -m pytest tests/test_extractor_gpu.py tests/test_stereo_gpu.py tests/test_lba_gpu.py -m gpu -x -q 2>&1 | tail -3
for v in 0 1 0 1; do echo "tum AOS2_OCT_PAIR=$v:"; AOS2_OCT_PAIR=$v python tools/prof_extract.py 512 2>&1 | tail -1 | cut -c1-330; done
for v in 0 1 0 1; do echo "kitti AOS2_OCT_PAIR=$v:"; AOS2_OCT_PAIR=$v python tools/prof_extract.py 256 kitti 2>&1 | tail -1 | cut -c1-330; done
python tools/gpu_fuzz_extractor.py 100000 100 2>&1 | tail -1
run() { name=$1; shift; env "$@" python bench.py --no-extra --no-cpu-baseline --steps 60 --warmup 6 $BA 2>$O/$name.err | tail -1 > $O/$name.json
  python - <<PY
import json
d = json.loads(open("$O/$name.json").read()); t = d["extra"]["timed_steps"]
print("%-12s %8.0f frames/s  %.3f ms/step  parity %s waits %s" % ("$name", d["value"], d["ms_per_step"], d["parity_checked"]["ok"], t["host_thread_waits_ms_per_step"]))
PY
}
run pair X=1
run nopair AOS2_OCT_PAIR=0
run pair2 X=1
run nopair2 AOS2_OCT_PAIR=0
BA="--workload kitti"
run kitti_pair X=1
run kitti_nopair AOS2_OCT_PAIR=0
run kitti_pair2 X=1
run kitti_nopair2 AOS2_OCT_PAIR=0

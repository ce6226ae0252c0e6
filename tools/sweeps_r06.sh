# Round 6 A/B sweeps, one gpurun call per section: bash tools/sweeps_r06.sh <section>   (results: profiles/r06_composite_decomposition.txt)
#   a  composite with a part switched off / stream priorities      b  wave priority of the LocalBA kernels (variant libs: tools/build_variant_lib.sh)
#   c  reduced-system kernels of round 4, hardware queues          d  k_schur windows dealt by cost                 e  LocalBA windows of k steps per call
#   f  one-workgroup-per-image octree                               g  octree pairs A/B + parity                     h, i  FAST pre-test variants      j  walk chunk sizes
sec_a() {
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
}
sec_b() {
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
}
sec_c() {
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
}
sec_d() {
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_lba_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
echo "deal by units (round 5):"; AOS2_LBA_DEAL_BY_UNITS=1 python tools/gpu_lba_mix_prof.py 2>&1 | tail -2
echo "deal by cost:"; python tools/gpu_lba_mix_prof.py 2>&1 | tail -2
done
echo "hom:"; LBA_MIX=hom AOS2_LBA_DEAL_BY_UNITS=1 python tools/gpu_lba_mix_prof.py 2>&1 | tail -1; LBA_MIX=hom python tools/gpu_lba_mix_prof.py 2>&1 | tail -1
echo "32:"; LBA_N=32 AOS2_LBA_DEAL_BY_UNITS=1 python tools/gpu_lba_mix_prof.py 2>&1 | tail -1; LBA_N=32 python tools/gpu_lba_mix_prof.py 2>&1 | tail -1
}
sec_e() {
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
}
sec_f() {
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
}
sec_g() {
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
}
sec_h() {
cd $GRAFT_REPO_ROOT; L=$PWD/active-orb-slam2_amd/lib
for v in fastma; do echo "tests with $v:"; AOS2_LIB=$L/libaos2_$v.so python -m pytest tests/test_extractor_gpu.py -m gpu -x -q 2>&1 | tail -2; AOS2_LIB=$L/libaos2_$v.so python tools/gpu_fuzz_extractor.py 100000 60 2>&1 | tail -1; done
for rep in 1 2 3; do for v in "" fastm fasta fastma; do lib=$L/libaos2${v:+_$v}.so; echo -n "${v:-base}: "; AOS2_LIB=$lib python tools/prof_extract.py 512 2>&1 | tail -1 | sed 's/.*fast_ms/fast_ms/'; done; done
}
sec_i() {
cd $GRAFT_REPO_ROOT; L=$PWD/active-orb-slam2_amd/lib
python -m pytest tests/test_extractor_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -2
python tools/gpu_fuzz_extractor.py 100000 90 2>&1 | tail -1
for rep in 1 2 3 4; do for v in nomargin ""; do lib=$L/libaos2${v:+_$v}.so; echo -n "${v:-margin}: "; AOS2_LIB=$lib python tools/prof_extract.py 512 2>&1 | tail -1 | sed 's/.*fast_ms/fast_ms/'; done; done
}
sec_j() {
cd $GRAFT_REPO_ROOT; L=$PWD/active-orb-slam2_amd/lib
for rep in 1 2 3; do for v in "" wc6 wc8; do lib=$L/libaos2${v:+_$v}.so; echo -n "${v:-wc4}: "; AOS2_LIB=$lib python tools/gpu_lba_mix_prof.py 2>&1 | tail -1; done; done
AOS2_LIB=$L/libaos2_wc6.so python -m pytest tests/test_lba_gpu.py -m gpu -x -q 2>&1 | tail -2
}
case "$1" in a|b|c|d|e|f|g|h|i|j) sec_$1;; *) echo "usage: bash tools/sweeps_r06.sh <a..j>";; esac

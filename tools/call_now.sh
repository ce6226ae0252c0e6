cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3
timeout 900 python -m pytest tests/test_lba_gpu.py tests/test_frames_gpu.py -m gpu -x -q -k "pose or frames or chain or Pose" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_bench_gpu.py -m gpu -x -q 2>&1 | tail -5
python tools/gpu_chain_latency.py 2>&1 | grep -v amdgpu.ids | head -3
AOS2_LIB=$GRAFT_REPO_ROOT/active-orb-slam2_amd/lib/libaos2_potiming.so python tools/gpu_chain_latency.py 2>&1 | grep "^PO n" | sort | uniq -c | sort -rn | head -6
for cfg in "8 2" "8 1" "4 2" "4 1" "8 2" "8 1"; do
  set -- $cfg
  echo "GPU_MAX_HW_QUEUES=$1 AOS2_LBA_GROUPS=$2"
  GPU_MAX_HW_QUEUES=$1 AOS2_LBA_GROUPS=$2 python bench.py --no-extra --no-cpu-baseline --no-verify 2> gpurun_out/c3/hwq.err | tail -1 > gpurun_out/c3/hwq_$1_$2.json
  python -c "
import json; d=json.loads(open('gpurun_out/c3/hwq_$1_$2.json').read()); print('   value %.0f ms_per_step %.3f' % (d['value'], d['ms_per_step'])); print({k: (round(v,2) if isinstance(v,float) else None) for k,v in d['stage_ms'].items()})" || tail -3 gpurun_out/c3/hwq.err
done

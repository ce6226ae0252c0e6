cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_frames_gpu.py -m gpu -x -q 2>&1 | tail -3
for g in 64 16 8 4 2 1 default; do if [ $g = default ]; then python tools/gpu_fill_group_sweep.py 2>&1 | grep FILL; else AOS2_FILL_GROUP=$g python tools/gpu_fill_group_sweep.py 2>&1 | grep FILL; fi; done
CHAIN_B=256 python tools/gpu_fill_group_sweep.py 2>&1 | grep FILL
CHAIN_B=256 AOS2_FILL_GROUP=64 python tools/gpu_fill_group_sweep.py 2>&1 | grep FILL

cd $GRAFT_REPO_ROOT; L=$PWD/active-orb-slam2_amd/lib
for rep in 1 2 3; do for v in "" wc6 wc8; do lib=$L/libaos2${v:+_$v}.so; echo -n "${v:-wc4}: "; AOS2_LIB=$lib python tools/gpu_lba_mix_prof.py 2>&1 | tail -1; done; done
AOS2_LIB=$L/libaos2_wc6.so python -m pytest tests/test_lba_gpu.py -m gpu -x -q 2>&1 | tail -2

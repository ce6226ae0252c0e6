cd $GRAFT_REPO_ROOT; L=$PWD/active-orb-slam2_amd/lib
python -m pytest tests/test_extractor_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -2
python tools/gpu_fuzz_extractor.py 100000 90 2>&1 | tail -1
for rep in 1 2 3 4; do for v in nomargin ""; do lib=$L/libaos2${v:+_$v}.so; echo -n "${v:-margin}: "; AOS2_LIB=$lib python tools/prof_extract.py 512 2>&1 | tail -1 | sed 's/.*fast_ms/fast_ms/'; done; done

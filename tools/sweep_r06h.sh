cd $GRAFT_REPO_ROOT; L=$PWD/active-orb-slam2_amd/lib
for v in fastma; do echo "tests with $v:"; AOS2_LIB=$L/libaos2_$v.so python -m pytest tests/test_extractor_gpu.py -m gpu -x -q 2>&1 | tail -2; AOS2_LIB=$L/libaos2_$v.so python tools/gpu_fuzz_extractor.py 100000 60 2>&1 | tail -1; done
for rep in 1 2 3; do for v in "" fastm fasta fastma; do lib=$L/libaos2${v:+_$v}.so; echo -n "${v:-base}: "; AOS2_LIB=$lib python tools/prof_extract.py 512 2>&1 | tail -1 | sed 's/.*fast_ms/fast_ms/'; done; done

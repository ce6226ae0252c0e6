# A/B of LocalBA variants: tests on the default build, then the batch timings (tools/gpu_lba_mix_prof.py) per setting.
#   VARIANTS="name ..." -> lib/libaos2_<name>.so (tools/build_lba_abl_libs.sh);  ENVS="VAR=val ..." -> environment settings, one run each
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_lba_gpu.py -x -q 2>&1 | tail -5
run() {
  for m in ${MIXES:-het hom}; do LBA_MIX=$m python tools/gpu_lba_mix_prof.py 2>&1 | grep windows | tail -1; done
  LBA_N=32 LBA_MIX=hom python tools/gpu_lba_mix_prof.py 2>&1 | grep windows | tail -1
}
echo "=== default"; run
for v in ${VARIANTS}; do echo "=== lib $v"; AOS2_LIB=$GRAFT_REPO_ROOT/active-orb-slam2_amd/lib/libaos2_$v.so run; done
for e in ${ENVS}; do echo "=== env $e"; env $e bash -c "$(declare -f run); run"; done

# A/B of LocalBA kernel variants (lib/libaos2_<name>.so from tools/build_lba_abl_libs.sh): tests on the default build, then kernel stats per variant
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_lba_gpu.py -x -q 2>&1 | tail -5
for v in "" ${VARIANTS}; do
  if [ -n "$v" ]; then export AOS2_LIB=$GRAFT_REPO_ROOT/active-orb-slam2_amd/lib/libaos2_$v.so; fi
  echo "=== variant '${v:-default}'"
  MIXES="${MIXES:-het hom}" bash tools/prof_lba_mix.sh 2>&1 | grep -v "^$" | grep -E "windows|k_schur|k_lin|k_points|k_ldlt" | awk 'NR%7!=2 && NR%7!=3'
done
unset AOS2_LIB
cd /tmp; LBA_N=32 LBA_MIX=hom python $GRAFT_REPO_ROOT/tools/gpu_lba_mix_prof.py 2>&1 | tail -2

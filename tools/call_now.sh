cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lba_gpu.py -m gpu -x -q 2>&1 | tail -4
for g in "" 1; do
  for cfg in "het 64" "hom 64" "hom 32"; do
    set -- $cfg
    echo "AOS2_LBA_GROUPS=${g:-default} $(env ${g:+AOS2_LBA_GROUPS=$g} LBA_MIX=$1 LBA_N=$2 python tools/gpu_lba_mix_prof.py 2>&1 | grep windows | tail -1)"
  done
done

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_lba_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
echo "deal by units (round 5):"; AOS2_LBA_DEAL_BY_UNITS=1 python tools/gpu_lba_mix_prof.py 2>&1 | tail -2
echo "deal by cost:"; python tools/gpu_lba_mix_prof.py 2>&1 | tail -2
done
echo "hom:"; LBA_MIX=hom AOS2_LBA_DEAL_BY_UNITS=1 python tools/gpu_lba_mix_prof.py 2>&1 | tail -1; LBA_MIX=hom python tools/gpu_lba_mix_prof.py 2>&1 | tail -1
echo "32:"; LBA_N=32 AOS2_LBA_DEAL_BY_UNITS=1 python tools/gpu_lba_mix_prof.py 2>&1 | tail -1; LBA_N=32 python tools/gpu_lba_mix_prof.py 2>&1 | tail -1

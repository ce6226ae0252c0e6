cd $GRAFT_REPO_ROOT
run() {
  for cfg in "het 64" "hom 64" "hom 32"; do set -- $cfg; echo "$V G=${G:-default} $(env ${G:+AOS2_LBA_GROUPS=$G} LBA_MIX=$1 LBA_N=$2 python tools/gpu_lba_mix_prof.py 2>&1 | grep windows | tail -1)"; done
}
V=base; G=; run; G=1; run
for v in wpe2 fma; do export AOS2_LIB=$GRAFT_REPO_ROOT/active-orb-slam2_amd/lib/libaos2_$v.so; V=$v; G=; run; G=1; run; done

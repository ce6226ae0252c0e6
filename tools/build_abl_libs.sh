#!/bin/bash
# variant builds of extractor_kernels.hip for A/B timing: tools/build_abl_libs.sh name "-Dflags" [name "-Dflags" ...]
# -> lib/libaos2_<name>.so
set -e
cd "$(dirname "$0")/../active-orb-slam2_amd/csrc"
mkdir -p /tmp/ablbuild
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics"
for f in extractor matcher lba pose_opt stereo vocabulary debug_taps; do
  [ /tmp/ablbuild/$f.o -nt $f.hip ] || /opt/rocm/bin/hipcc $F -c $f.hip -o /tmp/ablbuild/$f.o &
done
args=("$@")
for ((i=0;i<${#args[@]};i+=2)); do
  /opt/rocm/bin/hipcc $F ${args[i+1]} -Rpass-analysis=kernel-resource-usage -c extractor_kernels.hip -o /tmp/ablbuild/ek_${args[i]}.o 2> /tmp/ablbuild/ek_${args[i]}.log &
done
wait
for ((i=0;i<${#args[@]};i+=2)); do
  n=${args[i]}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libaos2_$n.so /tmp/ablbuild/ek_$n.o /tmp/ablbuild/{extractor,matcher,lba,pose_opt,stereo,vocabulary,debug_taps}.o -lpthread -ldl
  echo "$n: $(grep -A8 'Name: _ZN4aos215describe' /tmp/ablbuild/ek_$n.log | grep -E ' VGPRs:|Scratch|Occupancy' | sed 's/.*remark: *//' | tr '\n' ' ')"
done

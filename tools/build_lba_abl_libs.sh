#!/bin/bash
# variant builds of lba.hip for A/B timing: tools/build_lba_abl_libs.sh name "-Dflags" [name "-Dflags" ...] -> lib/libaos2_<name>.so
set -e
cd "$(dirname "$0")/../active-orb-slam2_amd/csrc"
make -s
mkdir -p /tmp/ablbuild
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics"
args=("$@")
for ((i=0;i<${#args[@]};i+=2)); do
  /opt/rocm/bin/hipcc $F ${args[i+1]} -c lba.hip -o /tmp/ablbuild/lba_${args[i]}.o &
done
wait
for ((i=0;i<${#args[@]};i+=2)); do
  n=${args[i]}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libaos2_$n.so /tmp/ablbuild/lba_$n.o ../build/{extractor_kernels,extractor,matcher,pose_opt,stereo,vocabulary,debug_taps}.o -lpthread -ldl
done

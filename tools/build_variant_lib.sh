#!/bin/bash
# variant build of ONE translation unit for A/B timing: tools/build_variant_lib.sh <unit without .hip> <name> "<-D flags>" [<name> "<flags>" ...]
#   -> active-orb-slam2_amd/lib/libaos2_<name>.so   (choose it with AOS2_LIB=<path>)
set -e
cd "$(dirname "$0")/../active-orb-slam2_amd/csrc"
make -s -j8
unit=$1; shift
mkdir -p /tmp/varbuild
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics"
args=("$@")
for ((i=0;i<${#args[@]};i+=2)); do
  /opt/rocm/bin/hipcc $F ${args[i+1]} -c $unit.hip -o /tmp/varbuild/${unit}_${args[i]}.o &
done
wait
for ((i=0;i<${#args[@]};i+=2)); do
  n=${args[i]}
  objs=""
  for u in extractor_kernels extractor matcher lba pose_opt stereo vocabulary replay debug_taps; do
    if [ $u = $unit ]; then objs="$objs /tmp/varbuild/${unit}_$n.o"; else objs="$objs ../build/$u.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libaos2_$n.so $objs -lpthread -ldl
done
ls -la ../lib

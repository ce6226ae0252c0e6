#!/bin/bash
# builds lib/libaos2_prof.so = the library with -DAOS2_OCT_PROF (octree phase counters, tools/gpu_oct_prof.py)
set -e
cd "$(dirname "$0")/../active-orb-slam2_amd/csrc"
mkdir -p /tmp/profbuild
for f in extractor_kernels extractor matcher lba pose_opt stereo vocabulary debug_taps; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -DAOS2_OCT_PROF -c $f.hip -o /tmp/profbuild/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libaos2_prof.so /tmp/profbuild/*.o -lpthread -ldl

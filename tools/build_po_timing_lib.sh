#!/bin/bash
# builds lib/libaos2_potiming.so = the library with -DAOS2_PO_TIMING (phase cycle counters of pose_optimization_kernel,
# printed by workgroup 0); use with AOS2_LIB=.../libaos2_potiming.so
set -e
cd "$(dirname "$0")/../active-orb-slam2_amd/csrc"
mkdir -p /tmp/pobuild
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -DAOS2_PO_TIMING -c pose_opt.hip -o /tmp/pobuild/pose_opt.o
objs=$(ls ../build/*.o | grep -v pose_opt.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libaos2_potiming.so $objs /tmp/pobuild/pose_opt.o -lpthread -ldl

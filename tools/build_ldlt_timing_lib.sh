#!/bin/bash
# builds lib/libaos2_ldlttiming.so = the library with -DAOS2_LDLT_TIMING (phase cycle counters of k_ldlt_lds, printed with
# AOS2_LBA_TRACE=1); use with AOS2_LIB=.../libaos2_ldlttiming.so
set -e
cd "$(dirname "$0")/../active-orb-slam2_amd/csrc"
mkdir -p /tmp/ldltbuild
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -DAOS2_LDLT_TIMING -c lba.hip -o /tmp/ldltbuild/lba.o
objs=$(ls ../build/*.o | grep -v "/lba.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libaos2_ldlttiming.so $objs /tmp/ldltbuild/lba.o -lpthread -ldl

#!/bin/bash
# A/B builds: tools/build_variant.sh NAME "-DFLAG ..." [source without .hip, default ba_sweep] -> vdo_slam_amd/libvdo_hip_NAME.so = the product library with that
# one source compiled with the extra flags (load it with VDO_HIP_LIB=...; the product library is not touched)
set -e
cd "$(dirname "$0")/../vdo_slam_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -Wno-unused-value"
F=${3:-ba_sweep}
/opt/rocm/bin/hipcc $FLAGS $2 -c $F.hip -o /tmp/${F}_$1.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "k_sweep_tileILb1ELb1" | grep -i "VGPRs:\|Spill\|LDS\|Occupancy" | head -6
OBJS=$(ls *.o | grep -v "^$F.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvdo_hip_$1.so $OBJS /tmp/${F}_$1.o -ldl

#!/bin/bash
# A/B builds of the WHOLE library with extra flags: tools/build_variant_all.sh NAME "-DFLAG ..." -> vdo_slam_amd/libvdo_hip_NAME.so
# (load it with VDO_HIP_LIB=...; the product library is not touched).  For flags that change the host side and the kernels together (VDO_TILE_EPT).
set -e
cd "$(dirname "$0")/../vdo_slam_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -Wno-unused-value"
mkdir -p /tmp/objs_$1
for f in *.hip; do /opt/rocm/bin/hipcc $FLAGS $2 -c $f -o /tmp/objs_$1/${f%.hip}.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvdo_hip_$1.so /tmp/objs_$1/*.o -ldl
ls -la ../libvdo_hip_$1.so

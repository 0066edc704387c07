#!/bin/bash
# The g2o-operation-order build of the product library: cam_point / chi2_w3 as a product and then a sum, Huber by the IEEE square root and
# division (se3_dev.hpp: -DVDO_UNFUSED_CAMPOINT -DVDO_SLOW_HUBER) in every source that uses them -> vdo_slam_amd/libvdo_hip_g2o_order.so.
# tests/test_ba_gpu.py::test_g2o_operation_order_build_keeps_every_block_at_1e12 builds it on the GPU box and pins ALL its blocks - right-hand sides
# included - to the oracle at 1e-12 of their own largest entry; it is never shipped (.gpurunignore) and nothing else loads it (VDO_HIP_LIB selects it).
set -e
cd "$(dirname "$0")/../vdo_slam_amd/csrc"
SRCS="ba_sweep ba_solve ba_hub capi_ba"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -Xarch_host -march=${HOST_ARCH:-x86-64-v3} -Wno-unused-value -Wno-unused-function"
mkdir -p g2o_order
for f in $SRCS; do /opt/rocm/bin/hipcc $FLAGS -DVDO_UNFUSED_CAMPOINT -DVDO_SLOW_HUBER -c $f.hip -o g2o_order/$f.o & done
wait
for f in $SRCS; do test -s g2o_order/$f.o; done
REST=""
for s in *.hip; do
  b=${s%.hip}
  case " $SRCS " in *" $b "*) continue;; esac
  test -s $b.o || /opt/rocm/bin/hipcc $FLAGS -c $s -o $b.o        # (the product objects travel with the snapshot; a fresh checkout compiles them here)
  REST="$REST $b.o"
done
OBJS=""
for f in $SRCS; do OBJS="$OBJS g2o_order/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvdo_hip_g2o_order.so $OBJS $REST -ldl
echo ../libvdo_hip_g2o_order.so

#!/bin/bash
# g2o-operation-order build: parity of every block class at 1e-12 (test + the raw deviations of both libraries)
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ba_gpu.py -x -q -m gpu -k "g2o_operation_order" > gpurun_out/ay_test.log 2>&1; tail -5 gpurun_out/ay_test.log
VDO_HIP_LIB=/root/repo/vdo_slam_amd/libvdo_hip_g2o_order.so timeout 300 python tests/g2o_order_worker.py 2>&1 | grep G2O_ORDER | tee gpurun_out/ay_g2o.txt
timeout 300 python tests/g2o_order_worker.py 2>&1 | grep G2O_ORDER | tee gpurun_out/ay_product.txt

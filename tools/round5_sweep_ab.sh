#!/bin/bash
# round 5, GPU: A/B of the K18 sweep variants (tools/build_variant_all.sh: old = unfused c + IEEE Huber, hub = fast Huber only, fus = fused c only,
# default = both) + the BA parity tests on the default library.  usage (gpurun): bash tools/round5_sweep_ab.sh
R=$(pwd); O=gpurun_out/r05a; mkdir -p $O
for v in old hub fus; do
  echo "== $v"; VDO_HIP_LIB=$R/vdo_slam_amd/libvdo_hip_$v.so timeout 300 python tools/sweep_repeat_probe.py 2200000 2>&1 | tail -7
done > $O/ab.log 2>&1
echo "== default" >> $O/ab.log; timeout 300 python tools/sweep_repeat_probe.py 2200000 2>&1 | tail -7 >> $O/ab.log
cat $O/ab.log
timeout 1500 python -m pytest tests/test_ba_gpu.py tests/test_g2o_replay_gpu.py tests/test_host_classes_gpu.py tests/test_track_to_batch_gpu.py tests/test_dense_check_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -30 > $O/ba_tests.log; cat $O/ba_tests.log

#!/bin/bash
# round 5, GPU: the bank-aware edge placement (capi_ba.hip close_tile, VDO_BA_PLACE=0 switches it off) and the point planes in the sweep's LDS against the
# round-4 kernel (libvdo_hip_old.so) on one box + the BA parity tests on the default library.  usage (gpurun): bash tools/round5_sweep_ab2.sh
R=$(pwd); O=gpurun_out/r05c; mkdir -p $O
export VDO_BA_TILE_STATS=1
{ echo "== old library (round 4: unfused c, IEEE Huber, AoS points, pose-sorted placement)"; VDO_HIP_LIB=$R/vdo_slam_amd/libvdo_hip_old.so timeout 300 python tools/sweep_repeat_probe.py 2200000 2>&1 | tail -7

  echo "== new library (default)"; timeout 300 python tools/sweep_repeat_probe.py 2200000 2>&1 | tail -8; } > $O/ab.log 2>&1
cat $O/ab.log
unset VDO_BA_TILE_STATS
timeout 1500 python -m pytest tests/test_ba_gpu.py tests/test_g2o_replay_gpu.py tests/test_host_classes_gpu.py tests/test_track_to_batch_gpu.py tests/test_dense_check_gpu.py -m gpu -q --tb=short 2>&1 | tail -30 > $O/ba_tests.log; cat $O/ba_tests.log

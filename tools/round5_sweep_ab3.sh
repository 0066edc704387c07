#!/bin/bash
# round 5, GPU: non-temporal hints + next-tile prefetch in the sweep, A/B on one box.  usage (gpurun): bash tools/round5_sweep_ab3.sh
R=$(pwd); O=gpurun_out/r05d; mkdir -p $O
{ for v in old base5 nont nopf; do echo "== $v"; VDO_HIP_LIB=$R/vdo_slam_amd/libvdo_hip_$v.so timeout 300 python tools/sweep_repeat_probe.py 2200000 2>&1 | tail -6; done
  echo "== default (nt + prefetch)"; timeout 300 python tools/sweep_repeat_probe.py 2200000 2>&1 | tail -6
  echo "== base5 again"; VDO_HIP_LIB=$R/vdo_slam_amd/libvdo_hip_base5.so timeout 300 python tools/sweep_repeat_probe.py 2200000 2>&1 | tail -6; } > $O/ab.log 2>&1
cat $O/ab.log
timeout 600 python -m pytest tests/test_ba_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -5

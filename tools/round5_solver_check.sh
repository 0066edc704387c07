#!/bin/bash
# round 5, GPU: k_pcg_chain with the boundary matrices in LDS + k_schur_tile with plane layouts - BA tests, sharded tests, phase probe, ms per LM iteration
# usage (gpurun): bash tools/round5_solver_check.sh [tag]
O=gpurun_out/r05${1:-k}; mkdir -p $O
for combo in "" "VDO_BA_NO_TWIST=1" "VDO_BA_CHAIN_WAVES=1"; do
  echo "=== tests [$combo]"
  env $combo timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_edge_cases_gpu.py tests/test_g2o_replay_gpu.py -x -q -m gpu 2>&1 | tail -3
done
echo "=== test_dist / track_to_batch / host classes"; timeout 600 python -m pytest tests/test_dist.py tests/test_track_to_batch_gpu.py tests/test_host_classes_gpu.py -x -q -m gpu 2>&1 | tail -3
for n in large bench; do VDO_HIP_LIB=$PWD/vdo_slam_amd/libvdo_hip_pcgprof.so timeout 250 python tools/pcg_chain_phase_probe.py $n 2>&1 | tail -12; done | tee $O/pcg_probe.txt
timeout 600 python tools/ba_variant_probe.py bench large roof 2>&1 | grep "ms/LM" | tee $O/ba_probe.txt

#!/bin/bash
# round 5, GPU: twisted pose chains + closed-form block inverse in k_pchain_factor - the BA tests in all four combinations of the two switches, then ms per LM iteration
# usage (gpurun): bash tools/round5_chain_check.sh [tag]
O=gpurun_out/r05${1:-h}; mkdir -p $O
for combo in "" "VDO_BA_PCHAIN_CLOSED=1" "VDO_BA_NO_TWIST=1" "VDO_BA_CHAIN_GLOBAL=1" "VDO_BA_CHAIN_WAVES=1"; do
  echo "=== tests [$combo]"
  env $combo timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_edge_cases_gpu.py tests/test_g2o_replay_gpu.py -x -q -m gpu 2>&1 | tail -4
done
echo "=== test_dist / track_to_batch (default)"; timeout 600 python -m pytest tests/test_dist.py tests/test_track_to_batch_gpu.py -x -q -m gpu 2>&1 | tail -3
for combo in "" "VDO_BA_PCHAIN_CLOSED=1" "VDO_BA_NO_TWIST=1" "VDO_BA_NO_TWIST=1 VDO_BA_PCHAIN_CLOSED=1"; do
  echo "=== probe [$combo]"
  env $combo timeout 600 python tools/ba_variant_probe.py bench large 2>&1 | grep "ms/LM"
done | tee $O/chain_ab.txt

#!/bin/bash
# round 5, GPU: object-chain inputs staged a frame ahead + outputs written into mapped pinned memory - tests, then A/B of the per-frame legs on one box
# usage (gpurun): bash tools/round5_chain_latency_ab.sh [tag]
O=gpurun_out/r05${1:-q}; mkdir -p $O
timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_pipeline_gpu.py tests/test_track_sequence_gpu.py tests/test_system_gpu.py tests/test_host_classes_gpu.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2; do
for combo in "" "VDO_PIPE_NO_CHAIN_PRESTAGE=1 VDO_ARENA_NO_MAPPED_OUT=1" "VDO_PIPE_NO_CHAIN_PRESTAGE=1" "VDO_ARENA_NO_MAPPED_OUT=1"; do
  env $combo python bench.py --steps 60 --no-batch --no-cpu-baseline --no-host-inputs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$combo]', 'value %.1f deferred %.1f' % (d['value'], d['value_deferred']), d['config'].get('step_ms_p50_p90_max'))"
done; done | tee $O/chain_latency_ab.txt

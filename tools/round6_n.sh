#!/bin/bash
O=gpurun_out/r06n; mkdir -p $O
timeout 900 python -m pytest tests/test_track_sequence_gpu.py tests/test_system_gpu.py tests/test_pipeline_gpu.py tests/test_bench_sequence_gpu.py tests/test_windowed_ba_gpu.py tests/test_track_to_batch_gpu.py -q -x 2>&1 | tail -5 | tee $O/tests.log
run() { timeout 600 python bench.py --steps 20 --warmup 5 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-batch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 value', round(d['value'],1), 'deferred', round(d['value_deferred'],1), d['config']['step_ms_p50_p90_max'])"; }
for i in 1 2 3; do
  run new
  VDO_PIPE_NO_EARLY_CAM=1 run no_early_cam
  VDO_PIPE_NO_K9_SPLIT=1 run no_k9_split
  VDO_PIPE_NO_EARLY_CAM=1 VDO_PIPE_NO_K9_SPLIT=1 run old
done | tee $O/bench_ab.txt

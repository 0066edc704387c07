#!/bin/bash
O=gpurun_out/r06ak; mkdir -p $O
run() { timeout 600 python bench.py --steps 20 --warmup 5 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-batch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 value', round(d['value'],1), 'deferred', round(d['value_deferred'],1), d['config']['step_ms_p50_p90_max'])"; }
for i in 1 2 3; do
  run default
  ROC_ACTIVE_WAIT_TIMEOUT=1000 run active_wait_1000us
  ROC_ACTIVE_WAIT_TIMEOUT=100 run active_wait_100us
done | tee $O/active_wait_ab.txt

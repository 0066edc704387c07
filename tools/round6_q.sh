#!/bin/bash
O=gpurun_out/r06q; mkdir -p $O
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_omd_gpu.py tests/test_edge_cases_gpu.py tests/test_dist.py tests/test_g2o_replay_gpu.py tests/test_host_classes_gpu.py -q -x 2>&1 | tail -5 | tee $O/tests.log
for i in 1 2; do timeout 900 python bench.py --steps 5 --warmup 2 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: (round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('ms_per_lm_iter')}, d['roofline'].get('linearize_frac_model'), d['roofline'].get('frac'))"; done | tee $O/bench_batch.txt

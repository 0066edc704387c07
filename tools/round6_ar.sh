#!/bin/bash
O=gpurun_out/r06ar; mkdir -p $O
timeout 1500 python -m pytest tests/test_ba_gpu.py tests/test_omd_gpu.py tests/test_edge_cases_gpu.py tests/test_dist.py tests/test_host_classes_gpu.py -q -x 2>&1 | tail -4 | tee $O/tests.log
for m in packed mixed packed mixed; do
if [ $m = mixed ]; then export VDO_BA_MIXED_ORDER=1; else unset VDO_BA_MIXED_ORDER; fi
timeout 900 python bench.py --steps 5 --warmup 2 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$m', {k: (round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('ms_per_lm_iter')}, 'sweep_ms', round(r['avg_launch_ms'],4), 'frac_model', round(r['frac_model'],3), 'lin_ms', round(r['linearize_ms'],4))"
done | tee $O/order_ab.txt

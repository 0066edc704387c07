#!/bin/bash
O=gpurun_out/r06x; mkdir -p $O
timeout 1200 python -m pytest tests/test_omd_gpu.py tests/test_ba_gpu.py tests/test_edge_cases_gpu.py tests/test_dist.py tests/test_host_classes_gpu.py tests/test_windowed_ba_gpu.py -q -x 2>&1 | tail -5 | tee $O/tests.log
for m in twosided serial; do
if [ $m = serial ]; then export VDO_BA_CHAIN_SERIAL=1; fi
timeout 900 python bench.py --steps 5 --warmup 2 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', {k: (round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('ms_per_lm_iter')})"
done | tee $O/chain_ab.txt

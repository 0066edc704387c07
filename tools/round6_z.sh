#!/bin/bash
O=gpurun_out/r06z; mkdir -p $O
for i in 1 2; do for m in twosided serial; do
if [ $m = serial ]; then export VDO_BA_CHAIN_SERIAL=1; else unset VDO_BA_CHAIN_SERIAL; fi
timeout 900 python bench.py --steps 5 --warmup 2 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', {k: (round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('ms_per_lm_iter')})"
done; done | tee $O/chain_ab.txt

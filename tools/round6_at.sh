#!/bin/bash
O=gpurun_out/r06at; mkdir -p $O
for sc in 100 125 150 200; do
VDO_BA_DYN_SLOT_SCALE=$sc timeout 900 python bench.py --steps 5 --warmup 2 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('scale $sc', {k: (round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('ms_per_lm_iter')}, 'sweep_ms', round(r['avg_launch_ms'],4))"
done | tee $O/slot_scale.txt

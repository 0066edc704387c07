#!/bin/bash
O=gpurun_out/r06ao; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/suite.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: (round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('ms_per_lm_iter')}, d['roofline']['frac'])" | tee $O/bench.txt

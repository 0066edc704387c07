#!/bin/bash
O=gpurun_out/r06g; mkdir -p $O
for f in 0 1; do echo "VDO_BA_LIN_FORK=$f"; VDO_BA_LIN_FORK=$f timeout 300 python tools/sweep_only.py 2200000 2>&1 | grep "ms_linearize"; VDO_BA_LIN_FORK=$f timeout 300 python tools/sweep_only.py 600000 2>&1 | grep "ms_linearize"; done | tee $O/lin_fork_ab.txt
timeout 300 python tools/schur_only.py 2200000 2>&1 | tail -3 | tee $O/schur_only.txt
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_omd_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee $O/ba_tests.log
for f in 0 1; do VDO_BA_LIN_FORK=$f timeout 600 python bench.py --steps 20 --warmup 5 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fork=$f', {k:round(v,4) for k,v in d.items() if k.startswith('ms_per_lm')}, 'lin_ms', d['roofline']['linearize_ms'], 'lin_frac_model', d['roofline']['linearize_frac_model'], 'solver', {k:d['roofline_solver'].get(k) for k in ('avg_launch_ms','frac_model','error')})
"; done | tee $O/bench_fork_ab.txt

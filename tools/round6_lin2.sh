#!/bin/bash
O=gpurun_out/r06h; mkdir -p $O
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_omd_gpu.py tests/test_g2o_replay_gpu.py tests/test_host_classes_gpu.py tests/test_golden_gpu.py tests/test_track_to_batch_gpu.py -m gpu -q -x 2>&1 | tail -8 | tee $O/ba_tests.log
for f in 1 0; do if [ $f = 1 ]; then export VDO_BA_LIN_UNFUSED=1; else unset VDO_BA_LIN_UNFUSED; fi; echo "UNFUSED=$f"; timeout 300 python tools/sweep_only.py 2200000 2>&1 | grep "ms_linearize"; timeout 600 python bench.py --steps 20 --warmup 5 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:round(v,4) for k,v in d.items() if k.startswith('ms_per_lm')}, 'lin_ms', d['roofline']['linearize_ms'], 'lin_frac_model', d['roofline']['linearize_frac_model'], 'sweep', d['roofline']['avg_launch_ms'])
"; done | tee $O/bench_fused_ab.txt

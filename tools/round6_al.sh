#!/bin/bash
O=gpurun_out/r06al; mkdir -p $O
timeout 1500 python -m pytest tests/test_ba_gpu.py tests/test_omd_gpu.py tests/test_edge_cases_gpu.py tests/test_dist.py tests/test_host_classes_gpu.py tests/test_windowed_ba_gpu.py tests/test_track_to_batch_gpu.py tests/test_g2o_replay_gpu.py tests/test_golden_gpu.py tests/test_dense_check_gpu.py -q -x 2>&1 | tail -5 | tee $O/tests.log
for i in 1 2; do for m in spec nospec; do
if [ $m = nospec ]; then export VDO_BA_NO_SPEC_LIN=1; else unset VDO_BA_NO_SPEC_LIN; fi
timeout 900 python bench.py --steps 20 --warmup 5 --no-parity --no-host-inputs --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', {k: (round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k.startswith('ms_per_lm_iter')}, 'windowed', round(d.get('value_with_windowed_ba',0),1))"
done; done | tee $O/spec_lin_ab.txt

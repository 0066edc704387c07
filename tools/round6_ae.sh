#!/bin/bash
O=gpurun_out/r06ae; mkdir -p $O
timeout 1200 python -m pytest tests/test_windowed_ba_gpu.py tests/test_track_to_batch_gpu.py tests/test_host_classes_gpu.py tests/test_ba_gpu.py tests/test_dense_check_gpu.py tests/test_edge_cases_gpu.py tests/test_g2o_replay_gpu.py -q -x 2>&1 | tail -4 | tee $O/tests.log
VDO_BATCH_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-host-inputs --no-parity --no-cpu-baseline --no-live-pmc > $O/bench.json 2> $O/bench.err
grep "batch\]" $O/bench.err | head -6
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); print("value_with_windowed_ba", d.get("value_with_windowed_ba")); print({k: round(v,4) for k,v in d.items() if k.startswith('ms_per_lm_iter')})
PY

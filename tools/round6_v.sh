#!/bin/bash
O=gpurun_out/r06v; mkdir -p $O
timeout 900 python -m pytest tests/test_windowed_ba_gpu.py tests/test_track_to_batch_gpu.py tests/test_host_classes_gpu.py tests/test_ba_gpu.py tests/test_g2o_replay_gpu.py tests/test_edge_cases_gpu.py -q -x 2>&1 | tail -4 | tee $O/tests.log
( time timeout 1200 python bench.py --steps 20 --warmup 5 --no-batch --no-host-inputs --no-parity > $O/bench.json 2> $O/bench.err ) 2> $O/time.txt; tail -3 $O/time.txt
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value","value_full_sequence","value_with_windowed_ba","speedup_vs_cpu_baseline"): print(k, d.get(k))
print(d["config"].get("value_with_windowed_ba")); print(d["cpu_baseline"].get("with_windowed_ba"), d["cpu_baseline"].get("with_windowed_ba_error"), d["config"].get("windowed_ba_cpu_sample_error"))
PY

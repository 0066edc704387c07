#!/bin/bash
O=gpurun_out/r06t; mkdir -p $O
( time timeout 1200 python bench.py --steps 20 --warmup 5 --no-batch --no-host-inputs --no-parity > $O/bench.json 2> $O/bench.err ) 2> $O/time.txt; tail -3 $O/time.txt; tail -3 $O/bench.err | cut -c1-400
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value","value_deferred","value_full_sequence","value_with_windowed_ba","speedup_vs_cpu_baseline"): print(k, d.get(k))
print(d["config"].get("value_with_windowed_ba")); print(d["cpu_baseline"].get("with_windowed_ba"), d["cpu_baseline"].get("with_windowed_ba_error"), d["config"].get("windowed_ba_cpu_sample_error"))
PY

#!/bin/bash
O=gpurun_out/r06af; mkdir -p $O
( time python bench.py --no-batch --no-host-inputs > $O/bench.json 2> $O/bench.err ) 2> $O/time.txt; tail -3 $O/time.txt; tail -2 $O/bench.err | cut -c1-300
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value","steps","value_full_sequence","value_with_windowed_ba","speedup_vs_cpu_baseline"): print(k, d.get(k))
print(d["cpu_baseline"].get("with_windowed_ba"), d["cpu_baseline"].get("with_windowed_ba_error"))
PY

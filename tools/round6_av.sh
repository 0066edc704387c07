#!/bin/bash
O=gpurun_out/r06av; mkdir -p $O
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]; c=d["cpu_baseline"]
print("value", d["value"], "x cpu", d["value"]/c["value"], "windowed", d.get("value_with_windowed_ba"), "host_sync", d.get("value_host_inputs_sync"), "cpu", c["value"])
print("roofline frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "lin_ms", r["linearize_ms"], "lin_frac_model", r["linearize_frac_model"])
print({k:round(v,4) for k,v in d.items() if k.startswith("ms_per_lm")})
PY

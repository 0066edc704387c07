#!/bin/bash
# round 5, GPU: the whole GPU test suite file by file, then the default bench line.  usage (gpurun): bash tools/round5_suite_bench.sh [tag]
O=gpurun_out/r05${1:-e}; mkdir -p $O
bash tools/gpu_suite_by_file.sh $O/suite.log > $O/suite_summary.txt 2>&1
tail -60 $O/suite_summary.txt
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt; tail -5 $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]; c=d["cpu_baseline"]
print("value", d["value"], "steps", d["steps"], "x cpu", d["value"]/c["value"], "host_sync", d.get("value_host_inputs_sync"), "cpu", c["value"])
print("roofline frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "traffic", r["traffic"], "lin_ms", r["linearize_ms"], "lin_frac_model", r["linearize_frac_model"], "valu", r.get("valu"), "lds", r.get("lds"))
print({k:v for k,v in d.items() if k.startswith("ms_per_lm")})
PY

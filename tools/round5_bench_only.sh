#!/bin/bash
# round 5: the two bench lines (default = 148 steps; the driver's window) on one box.  usage (gpurun): bash tools/round5_bench_only.sh [tag]
O=gpurun_out/r05${1:-b}; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
python - <<PY
import json
for f in ("bench.json", "bench_steps20.json"):
    d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); r=d["roofline"]; c=d["cpu_baseline"]
    print(f, "value", d["value"], "steps", d["steps"], "x cpu", d["value"]/c["value"], "deferred", d.get("value_deferred"), "host_sync", d.get("value_host_inputs_sync"), "cpu", c["value"], d["config"].get("step_ms_p50_p90_max"))
    print("  roofline frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "lin_ms", r["linearize_ms"], "lin_frac_model", r["linearize_frac_model"], {k:round(v,4) for k,v in d.items() if k.startswith("ms_per_lm")})
PY

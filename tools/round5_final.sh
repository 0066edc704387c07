#!/bin/bash
# round 5, the closing GPU pass: the whole GPU suite file by file, the default bench line (148 steps), the driver's window (--steps 20 --warmup 5), the two-rank line on
# the shared GPU, kernel statistics of the LM on the roofline graph.   usage (gpurun): bash tools/round5_final.sh [tag]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05${1:-y}; mkdir -p $O
bash tools/gpu_suite_by_file.sh $O/suite.log > $O/suite_summary.txt 2>&1; tail -48 $O/suite_summary.txt
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt; tail -3 $O/bench.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err ) 2> $O/bench_steps20_time.txt; tail -3 $O/bench_steps20_time.txt
python bench.py --gpus 2 --steps 20 --no-cpu-baseline --no-host-inputs > $O/bench_gpus2_shared.json 2> $O/bench_gpus2.err; tail -2 $O/bench_gpus2.err
python - <<PY
import json
for f in ("bench.json", "bench_steps20.json"):
    d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); r=d["roofline"]; c=d["cpu_baseline"]
    print(f, "value", d["value"], "steps", d["steps"], "x cpu", d["value"]/c["value"], "deferred", d.get("value_deferred"), "host_sync", d.get("value_host_inputs_sync"), "cpu", c["value"], d["config"].get("step_ms_p50_p90_max"))
    print("  roofline frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "traffic", r["traffic"], "lin_ms", r["linearize_ms"], "lin_frac_model", r["linearize_frac_model"], (r.get("valu") or {}).get("issue_frac"), (r.get("lds") or {}).get("busy_frac"))
    print("  ", {k:round(v,4) for k,v in d.items() if k.startswith("ms_per_lm")})
d=json.loads(open("$O/bench_gpus2_shared.json").read().strip().splitlines()[-1]); print("gpus2", d["value"], {k:(v["ms_per_lm_iter_sharded"], v["ms_per_lm_iter_1gpu"], v["same_trajectory_as_1gpu"]) for k,v in d["sharded"].items()})
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_ba_roof -- python $R/tools/ba_variant_probe.py roof > $O/ba_roof.log 2>&1
cd $R; python tools/rocprof_summary.py $(find $O/prof_ba_roof -name "*.db" | head -1) 40 > $O/ba_roof_kernel_stats.txt 2>&1; grep "ms/LM" $O/ba_roof.log; head -8 $O/ba_roof_kernel_stats.txt | cut -c1-140
find $O -name "*.db" -delete

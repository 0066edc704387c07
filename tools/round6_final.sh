#!/bin/bash
# round 6, the closing GPU pass: the GPU suite file by file, the default bench line (148 steps), the driver's window (--steps 20 --warmup 5), the two-rank line on the shared GPU,
# the K18 passes on the roofline graph (kernel statistics, the two HBM-traffic PMC passes, SQ counters), kernel statistics of the per-frame legs.
# usage (gpurun): bash tools/round6_final.sh [tag]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06${1:-final}; mkdir -p $O
bash tools/gpu_suite_by_file.sh $O/suite.log > $O/suite_summary.txt 2>&1; tail -50 $O/suite_summary.txt
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt; tail -3 $O/bench.err | cut -c1-300
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err ) 2> $O/bench_steps20_time.txt; tail -3 $O/bench_steps20_time.txt
python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-host-inputs --no-parity --no-full-sequence > $O/bench_gpus2_shared.json 2> $O/bench_gpus2.err; tail -2 $O/bench_gpus2.err | cut -c1-300
python - <<PY
import json
for f in ("bench.json", "bench_steps20.json"):
    d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); r=d["roofline"]; c=d["cpu_baseline"]
    print(f, "value", d["value"], "steps", d["steps"], "x cpu", d["value"]/c["value"], "deferred", d.get("value_deferred"), "full", d.get("value_full_sequence"), "windowed", d.get("value_with_windowed_ba"), "host_sync", d.get("value_host_inputs_sync"), "cpu", c["value"], d["config"].get("step_ms_p50_p90_max"))
    print("  speedups", d.get("speedup_vs_cpu_baseline"))
    print("  roofline frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "traffic", r["traffic"], "lin_ms", r["linearize_ms"], "lin_frac_model", r["linearize_frac_model"], (r.get("valu") or {}).get("issue_frac"), (r.get("lds") or {}).get("busy_frac"))
    print("  solver", d.get("roofline_solver"))
    print("  ", {k:round(v,4) for k,v in d.items() if k.startswith("ms_per_lm")})
    print("  parity", {k: v for k, v in (d.get("parity") or {}).items() if k in ("frames", "pose_bit_equal_frames", "index_sets_equal", "object_motions", "object_motions_within_1e-4", "first_divergence_frame")})
    print("  parity_full", {k: v for k, v in (d.get("parity_full_sequence") or {}).items() if k in ("frames", "pose_bit_equal_frames", "index_sets_equal", "object_motions", "object_motions_within_1e-4", "first_divergence_frame")})
d=json.loads(open("$O/bench_gpus2_shared.json").read().strip().splitlines()[-1]); print("gpus2", d["value"], {k:(v.get("ms_per_lm_iter_sharded"), v.get("ms_per_lm_iter_1gpu"), v.get("same_trajectory_as_1gpu"), v.get("allreduces_per_lm_iter")) for k,v in d["sharded"].items()})
PY
# K18 passes
cd /tmp && export TMPDIR=/tmp
N=2200000
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_sweep -- python $R/tools/sweep_only.py $N > $O/sweep.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -- python $R/tools/sweep_only.py $N > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -- python $R/tools/sweep_only.py $N > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_sq1 -- python $R/tools/sweep_only.py $N > $O/pmc_sq1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/pmc_sq2 -- python $R/tools/sweep_only.py $N > $O/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_frame -- python $R/bench.py --steps 60 --warmup 5 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-batch > $O/bench_frame_prof.json 2> $O/bench_frame_prof.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_ba_roof -- python $R/tools/ba_variant_probe.py roof > $O/ba_roof.log 2>&1
cd $R
python tools/rocprof_summary.py $(find $O/prof_sweep -name "*.db" | head -1) 40 > $O/sweep_kernel_stats.txt 2>&1
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
python tools/pmc_summary.py k_sweep_tile $F $W > $O/sweep_pmc_hbm_traffic.txt 2>&1
python tools/pmc_summary.py k_finalize_pose $F $W > $O/finalize_pmc_hbm_traffic.txt 2>&1
grep "^n_eb" $O/sweep.log | tail -1 >> $O/sweep_pmc_hbm_traffic.txt
python tools/pmc_counters.py k_sweep_tile $(find $O/pmc_sq1 -name "*.db" | head -1) $(find $O/pmc_sq2 -name "*.db" | head -1) > $O/sweep_sq_counters.txt 2>&1
python tools/rocprof_summary.py $(find $O/prof_frame -name "*.db" | head -1) 40 > $O/frame_kernel_stats.txt 2>&1
python tools/rocprof_summary.py $(find $O/prof_ba_roof -name "*.db" | head -1) 40 > $O/ba_roof_kernel_stats.txt 2>&1
find $O -name "*.db" -delete
cat $O/sweep_pmc_hbm_traffic.txt; head -6 $O/sweep_kernel_stats.txt | cut -c1-140; head -14 $O/sweep_sq_counters.txt; head -8 $O/frame_kernel_stats.txt | cut -c1-140

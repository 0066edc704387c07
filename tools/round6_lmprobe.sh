#!/bin/bash
O=gpurun_out/r06e; mkdir -p $O
timeout 900 python -m pytest tests/test_bench_sequence_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee $O/new_tests.log
bash tools/build_profiled_flow2.sh > $O/build_prof.log 2>&1; tail -2 $O/build_prof.log
VDO_HIP_LIB=$PWD/vdo_slam_amd/libvdo_hip_prof.so timeout 300 python tools/flow2_phase_probe.py 1200 o800 o400 o230 o120 > $O/flow2_phase_probe.txt 2>&1; cat $O/flow2_phase_probe.txt
rm -f vdo_slam_amd/libvdo_hip_prof.so
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2> $O/bench_driver_time.txt
tail -3 $O/bench_driver_time.txt; tail -5 $O/bench_driver.err
python - <<PY
import json
d=json.loads(open("$O/bench_driver.json").read().strip().splitlines()[-1])
c=d["cpu_baseline"]
print("value", d["value"], "x cpu", d["value"]/c["value"], "full", d.get("value_full_sequence"), d["speedup_vs_cpu_baseline"], "cpu", c["value"], c.get("full_sequence"))
for k in ("parity","parity_full_sequence"):
    p=d.get(k)
    if p: print(k, {q:p[q] for q in ("frames","pose_bit_equal_frames","index_sets_equal","first_divergence_frame","object_motions","object_motions_within_1e-4","object_motions_bit_equal","frames_equal_by_part","tracklets_equal","labels_outside")})
print(d["roofline"]["frac"], {k:v for k,v in d.items() if k.startswith("ms_per_lm")})
PY

#!/bin/bash
O=gpurun_out/r06d; mkdir -p $O
timeout 900 python -m pytest tests/test_bench_sequence_gpu.py tests/test_track_sequence_gpu.py -m gpu -q -x 2>&1 | tail -15 | tee $O/new_tests.log
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
timeout 300 python tools/step_events.py 60 > $O/step_events.txt 2>&1; tail -60 $O/step_events.txt

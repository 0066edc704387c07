#!/bin/bash
# round 5, GPU: does the device's performance level matter for the latency-bound per-frame path?  (and: the AVX2 host build)  usage (gpurun): bash tools/round5_perflevel_probe.sh
O=gpurun_out/r05g; mkdir -p $O
rocm-smi --showperflevel --showclocks 2>&1 | head -30 > $O/smi_before.txt
python bench.py --no-batch --no-cpu-baseline --no-host-inputs --steps 148 > $O/bench_auto.json 2> $O/bench_auto.err
rocm-smi --setperflevel high > $O/smi_set.txt 2>&1
rocm-smi --showperflevel --showclocks 2>&1 | head -30 > $O/smi_after.txt
python bench.py --no-batch --no-cpu-baseline --no-host-inputs --steps 148 > $O/bench_high.json 2> $O/bench_high.err
rocm-smi --setperflevel auto >> $O/smi_set.txt 2>&1
python bench.py --no-batch --no-cpu-baseline --no-host-inputs --steps 20 > $O/bench_auto20.json 2> $O/bench_auto20.err
python - <<PY
import json
for n in ("auto","high","auto20"):
    try:
        d=json.loads(open("$O/bench_%s.json"%n).read().strip().splitlines()[-1]); print(n, d["value"], d["value_deferred"], d["config"]["step_ms_p50_p90_max"], d["config"]["host_ms_per_section"])
    except Exception as e: print(n, "failed", e)
PY
cat $O/smi_set.txt | head; grep -i "perf\|sclk\|level" $O/smi_before.txt $O/smi_after.txt | head -20
timeout 900 python -m pytest tests/test_track_sequence_gpu.py tests/test_system_gpu.py tests/test_pipeline_gpu.py tests/test_host_classes_gpu.py tests/test_ransac_gpu.py -m gpu -q --tb=short 2>&1 | tail -4

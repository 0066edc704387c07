#!/bin/bash
cd /root/repo; O=gpurun_out/r06bf; mkdir -p $O
timeout 1500 python -m pytest tests/test_ba_gpu.py tests/test_windowed_ba_gpu.py tests/test_host_classes_gpu.py tests/test_golden_gpu.py tests/test_track_to_batch_gpu.py -x -q -m gpu > $O/t1.log 2>&1; tail -4 $O/t1.log
VDO_BATCH_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-host-inputs --no-parity > $O/bench.json 2> $O/bench.err
grep "vdo_ba_create\|^\[batch\]\|partial batch" $O/bench.err | tail -6 | cut -c1-260
python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1]); print(d["value"], d.get("value_full_sequence"), d.get("value_with_windowed_ba"))
PY

#!/bin/bash
cd /root/repo; O=gpurun_out/r06ci; mkdir -p $O
timeout 1200 python -m pytest tests/test_windowed_ba_gpu.py tests/test_host_classes_gpu.py tests/test_track_to_batch_gpu.py tests/test_system_gpu.py tests/test_tracking_gpu.py -x -q -m gpu > $O/t.log 2>&1; tail -3 $O/t.log
VDO_BATCH_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-host-inputs --no-parity > $O/b.json 2> $O/err.txt
grep "partial batch" $O/err.txt | tail -6 | cut -c1-170
python -c "
import json
d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1]); print(round(d['value'],1), round(d.get('value_full_sequence',0),1), round(d.get('value_with_windowed_ba',0),1))"

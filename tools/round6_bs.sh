#!/bin/bash
cd /root/repo; O=gpurun_out/r06bs; mkdir -p $O
for i in 1 2 3 4; do
for v in prio flat; do
  if [ $v = flat ]; then export VDO_FLAT_STREAM_PRIORITIES=1; else unset VDO_FLAT_STREAM_PRIORITIES; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-parity 2>$O/err_$v.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', round(d['value'],1), round(d.get('value_deferred',0),1), round(d.get('value_full_sequence',0),1), round(d.get('value_with_windowed_ba',0),1), round(d.get('value_host_inputs',0),1), round(d.get('value_host_inputs_sync',0),1), d['config'].get('step_ms_p50_p90_max'))" | tee -a $O/ab.txt
done; done

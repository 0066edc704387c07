#!/bin/bash
cd /root/repo; O=gpurun_out/r06bp; mkdir -p $O
for i in 1 2 3; do
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-host-inputs --no-parity 2>$O/err_$q.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('queues $q', round(d['value'],1), round(d.get('value_deferred',0),1), round(d.get('value_full_sequence',0),1), round(d.get('value_with_windowed_ba',0),1), d['config'].get('step_ms_p50_p90_max'))" | tee -a $O/ab.txt
done; done
GPU_MAX_HW_QUEUES=8 VDO_CHAIN_TRACE=1 timeout 600 python tools/step_events.py 20 > $O/events8.txt 2> $O/chain8.txt; grep "vdo_object_chain" $O/chain8.txt | tail -3; sed -n '/defer_objects=0/,/sections/p' $O/events8.txt | head -20

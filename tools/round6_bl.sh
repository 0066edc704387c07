#!/bin/bash
cd /root/repo; O=gpurun_out/r06bl; mkdir -p $O
for i in 1 2 3; do
for v in new old; do
  if [ $v = old ]; then export VDO_NO_TICKET_SYNC=1; else unset VDO_NO_TICKET_SYNC; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-host-inputs 2>$O/err_$v.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', round(d['value'],1), round(d.get('value_deferred',0),1), round(d.get('value_full_sequence',0),1), round(d.get('value_with_windowed_ba',0),1), d['config'].get('step_ms_p50_p90_max'), (d.get('parity') or {}).get('pose_bit_equal_frames'), (d.get('parity') or {}).get('index_sets_equal'))" | tee -a $O/ab.txt
done; done
unset VDO_NO_TICKET_SYNC
bash tools/gpu_suite_by_file.sh $O/suite.log > $O/suite_summary.txt 2>&1; tail -45 $O/suite_summary.txt

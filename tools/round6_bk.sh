#!/bin/bash
cd /root/repo; O=gpurun_out/r06bk; mkdir -p $O
timeout 1500 python -m pytest tests/test_ba_gpu.py tests/test_windowed_ba_gpu.py tests/test_host_classes_gpu.py tests/test_omd_gpu.py tests/test_g2o_replay_gpu.py -x -q -m gpu > $O/t1.log 2>&1; tail -3 $O/t1.log
for i in 1 2 3; do
for v in new old; do
  if [ $v = old ]; then export VDO_BA_NO_TICKET=1; else unset VDO_BA_NO_TICKET; fi
  VDO_BATCH_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-inputs --no-parity --no-live-pmc 2>$O/err_$v.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', round(d['value'],1), round(d.get('value_with_windowed_ba',0),1), {k:round(x,4) for k,x in d.items() if k.startswith('ms_per_lm')})" | tee -a $O/ab.txt
  grep "^\[batch\] P 20" $O/err_$v.txt | tail -2 | cut -c1-110
done; done

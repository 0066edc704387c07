#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r06bc; mkdir -p $O
cd $R
timeout 900 rocprofv3 --kernel-trace -d $O/trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-host-inputs --no-parity > $O/bench.txt 2> $O/bench.err
python tools/window_lm_timeline.py report $O/trace 5 > $O/report_5.txt 2>&1; cat $O/report_5.txt | tail -30
rm -rf $O/trace
for i in 1 2 3; do
for v in new old; do
  if [ $v = old ]; then export VDO_BA_NO_DENSE_SMALL=1; else unset VDO_BA_NO_DENSE_SMALL; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-host-inputs --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', round(d['value'],1), round(d.get('value_full_sequence',0),1), round(d.get('value_with_windowed_ba',0),1))" | tee -a $O/ab.txt
done; done

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r06bd; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ba_gpu.py -x -q -m gpu -k "small_reduced or dense_mfma or static_only or non_path" > $O/t1.log 2>&1; tail -4 $O/t1.log
timeout 900 python -m pytest tests/test_windowed_ba_gpu.py tests/test_host_classes_gpu.py -x -q -m gpu > $O/t2.log 2>&1; tail -3 $O/t2.log
timeout 900 rocprofv3 --kernel-trace -d $O/trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-host-inputs --no-parity > $O/bench.txt 2> $O/bench.err
python tools/window_lm_timeline.py report $O/trace 5 > $O/report_5.txt 2>&1; cat $O/report_5.txt | tail -30
rm -rf $O/trace
for i in 1 2; do
for v in c4 c2 c1; do
  export VDO_BA_DENSE_CHUNK=${v#c}
  VDO_BATCH_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-host-inputs --no-parity 2>$O/err_$v.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', round(d['value'],1), round(d.get('value_full_sequence',0),1), round(d.get('value_with_windowed_ba',0),1))" | tee -a $O/ab.txt
  grep "^\[batch\]" $O/err_$v.txt | tail -2 | cut -c1-120
done; done

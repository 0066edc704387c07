#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r06ba; mkdir -p $O
cd $R
VDO_BATCH_TRACE=1 timeout 900 rocprofv3 --kernel-trace -d $O/trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-host-inputs --no-parity > $O/bench.txt 2> $O/bench.err
grep -i "batch\|window" $O/bench.err | head -40
for w in 3 5 -1; do python tools/window_lm_timeline.py report $O/trace $w > $O/report_$w.txt 2>&1; head -3 $O/report_$w.txt; done
cat $O/report_5.txt | tail -40
rm -rf $O/trace

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r06bx; mkdir -p $O
cd $R
export VDO_PROBE_SHAPE=60,30000,5,800 VDO_PROBE_ITS=5
timeout 600 rocprofv3 --kernel-trace -d $O/trace -- python tools/window_lm_timeline.py run > $O/traced.txt 2>&1; tail -3 $O/traced.txt
python tools/window_lm_timeline.py report $O/trace > $O/report.txt 2>&1; cat $O/report.txt
rm -rf $O/trace

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r06bo; mkdir -p $O
cd $R
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace -- python tools/step_events.py 20 > $O/events.txt 2> $O/err.txt
python tools/frame_gpu_timeline.py $O/trace 40 > $O/timeline_40.txt 2>&1; cat $O/timeline_40.txt | head -90
rm -rf $O/trace

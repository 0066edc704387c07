#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06ad; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/tools/ba_probe.py 20 2300 0 0 27 0 > $O/probe.log 2>&1
cd $R; DB=$(find $O/prof -name "*.db" | head -1); python tools/rocprof_summary.py $DB 40 > $O/tiny_kernel_stats.txt 2>&1; tail -2 $O/probe.log; cut -c1-120 $O/tiny_kernel_stats.txt | head -40
find $O -name "*.db" -delete

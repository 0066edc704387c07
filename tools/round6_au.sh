#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06au; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_omd -- python $R/tools/ba_probe.py 300 150000 4 40000 5 0 > $O/probe_omd.log 2>&1
cd $R; python tools/rocprof_summary.py $(find $O/prof_omd -name "*.db" | head -1) 40 > $O/omd_kernel_stats.txt 2>&1; tail -1 $O/probe_omd.log; cut -c1-110 $O/omd_kernel_stats.txt | head -16
VDO_BA_TILE_STATS=1 python -c "
import sys; sys.path.insert(0,'$R')
from vdo_slam_amd import synth
from vdo_slam_amd.ba import BatchBA, Context
g = synth.make_ba_graph(300, 150000, 4, 40000, seed=1)
ba = BatchBA(Context(0), g); print(ba.dims())" 2>&1 | tail -2
find $O -name "*.db" -delete

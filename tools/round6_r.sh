#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for tag in omd large; do
  if [ $tag = omd ]; then A="300 150000 4 40000 5 0"; else A="239 950000 20 500 3 0"; fi
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -- python $R/tools/ba_probe.py $A > $O/probe_$tag.log 2>&1
done
cd $R
for tag in omd large; do DB=$(find $O/prof_$tag -name "*.db" | head -1); python tools/rocprof_summary.py $DB 40 > $O/${tag}_kernel_stats.txt 2>&1; tail -2 $O/probe_$tag.log; cut -c1-150 $O/${tag}_kernel_stats.txt | head -26; done
find $O -name "*.db" -delete

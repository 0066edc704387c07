#!/bin/bash
# round 5, GPU: product vs the whole reference (libref_full) + the chain-operator test, then kernel statistics of the LM on the 1 M-point and the roofline graph
# usage (gpurun): bash tools/round5_chain_profile.sh [tag]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05${1:-i}; mkdir -p $O
timeout 900 python -m pytest tests/test_system_gpu.py "tests/test_ba_gpu.py::test_pose_chain_solver_is_the_same_operator_however_it_is_partitioned" -x -q -m gpu 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
for n in large roof; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_ba_$n -- python $R/tools/ba_variant_probe.py $n > $O/ba_$n.log 2>&1
done
cd $R
for n in large roof; do DB=$(find $O/prof_ba_$n -name "*.db" | head -1); python tools/rocprof_summary.py $DB 40 > $O/ba_${n}_kernel_stats.txt 2>&1; grep "ms/LM" $O/ba_$n.log; cut -c1-150 $O/ba_${n}_kernel_stats.txt | head -24; done
find $O -name "*.db" -size +20M -delete

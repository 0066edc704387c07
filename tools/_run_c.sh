cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ba_gpu.py -x -q -m gpu 2>&1 | tail -5
for m in 0 1; do echo "chain scan $m"; VDO_BA_CHAIN_SCAN=$m python tools/ba_probe.py 200 600000 10 1500 3 0 2>&1 | tail -2; done
for m in 0 1; do echo "bench graph chain scan $m"; VDO_BA_CHAIN_SCAN=$m python tools/ba_probe.py 60 30000 5 800 5 0 2>&1 | tail -1; done
cd /tmp; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03c; mkdir -p $O
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_ba_large -- python $GRAFT_REPO_ROOT/tools/ba_probe.py 200 600000 10 1500 3 0 > $O/ba_large.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find $O/prof_ba_large -name "*.db" | head -1); python tools/rocprof_summary.py $DB 30 2>&1 | tee $O/ba_large_kernel_stats.txt; find $O -name "*.db" -size +20M -delete

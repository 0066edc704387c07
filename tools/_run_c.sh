cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ba_gpu.py tests/test_host_classes_gpu.py -x -q -m gpu 2>&1 | tail -2
for e in 0 1; do
if [ $e = 1 ]; then export VDO_BA_TILE_ORDER_IDENTITY=1; fi
echo "identity order: $e"
python tools/ba_probe.py 200 600000 10 1500 3 0 2>&1 | tail -1
python tools/ba_probe.py 60 30000 5 800 5 0 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03c; mkdir -p $O; rm -rf $O/prof_ba_large
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_ba_large -- python $GRAFT_REPO_ROOT/tools/ba_probe.py 200 600000 10 1500 3 0 > $O/ba_large.log 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find $O/prof_ba_large -name "*.db" | head -1); python tools/rocprof_summary.py $DB 20 2>&1 | grep "schur\|precond_tile"; find $O -name "*.db" -size +20M -delete
done

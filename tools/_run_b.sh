cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
python -m pytest tests/test_ba_gpu.py tests/test_g2o_replay_gpu.py tests/test_host_classes_gpu.py tests/test_track_to_batch_gpu.py tests/test_dist.py -x -q -m gpu 2>&1 | tail -8
python tools/sweep_only.py 2>&1 | tail -4
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof_sweep -- python $R/tools/sweep_only.py > $O/sweep.log 2>&1
cd $R
DB=$(find $O/prof_sweep -name "*.db" | head -1); python tools/rocprof_summary.py $DB 12 2>&1 | tee $O/sweep_kernel_stats.txt
find $O -name "*.db" -size +20M -delete

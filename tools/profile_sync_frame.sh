# Kernel + copy timeline of a few synchronous-mode frames of the bench sequence (run through gpurun from the repo root).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05p}
mkdir -p $O
VDO_BENCH_SYNC_OBJECTS=1 timeout 250 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof_sync -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-host-inputs --no-batch > $O/bench_sync_under_rocprof.json 2> $O/bench_sync_under_rocprof.err
cd $R
DB=$(find $O/prof_sync -name "*.db" | head -1)
python tools/rocprof_timeline.py $DB 60 2600 > $O/sync_frame_timeline.txt 2>&1
python tools/rocprof_summary.py $DB 40 > $O/sync_kernel_stats.txt 2>&1
find $O -name "*.db" -size +20M -delete
head -70 $O/sync_frame_timeline.txt

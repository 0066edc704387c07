cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof_bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/final/bench_under_rocprof.json 2> $R/gpurun_out/final/bench_under_rocprof.err
timeout 150 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof_ba_small -- python $R/tools/ba_probe.py 60 30000 5 800 5 0 > $R/gpurun_out/final/ba_small.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof_ba_large -- python $R/tools/ba_probe.py 200 600000 10 1500 3 0 > $R/gpurun_out/final/ba_large.log 2>&1
cd $R
for n in bench ba_small ba_large; do DB=$(find gpurun_out/final/prof_$n -name "*.db" | head -1); python tools/rocprof_summary.py $DB 30 > gpurun_out/final/${n}_kernel_stats.txt 2>&1; done
timeout 250 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
tail -c 600 gpurun_out/final/bench.json

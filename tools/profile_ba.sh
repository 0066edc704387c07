# rocprofv3 kernel statistics of the LM on the roofline graph (2 190 poses) and on the bench graph -> gpurun_out/r03e/ (run through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_ba_large -- python $R/tools/ba_probe.py 200 600000 10 1500 3 0 > $O/ba_large.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_ba_bench -- python $R/tools/ba_probe.py 60 30000 5 800 5 0 > $O/ba_bench.log 2>&1
cd $R
for n in ba_large ba_bench; do DB=$(find $O/prof_$n -name "*.db" | head -1); python tools/rocprof_summary.py $DB 40 > $O/${n}_kernel_stats.txt 2>&1; done
tail -3 $O/ba_large.log; cut -c1-150 $O/ba_large_kernel_stats.txt | head -24
find $O -name "*.db" -size +20M -delete

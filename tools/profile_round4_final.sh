# Round-4 closing pass (run through gpurun from the repo root): the K18 sweep passes (kernel statistics + PMC, tools/profile_round4_sweep.sh), the batch LM on three
# graph shapes (tools/profile_round4_ba.sh), rocprofv3 kernel statistics of the per-frame bench, then the bench itself with the fresh counter file in place.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
bash $R/tools/profile_round4_sweep.sh 2200000 > $O/profile_sweep.log 2>&1
cp $O/sweep_pmc_hbm_traffic.txt $R/profiles/r04_sweep_pmc_hbm_traffic.txt          # (bench.py quotes the counter-based traffic taken on this very layout)
bash $R/tools/profile_round4_ba.sh > $O/profile_ba.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats -d $O/prof_bench -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-host-inputs --no-batch > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $R
DB=$(find $O/prof_bench -name "*.db" | head -1); python tools/rocprof_summary.py $DB 60 > $O/bench_kernel_stats.txt 2>&1
timeout 600 python bench.py > $O/bench_final.json 2> $O/bench_final.err
find $O -name "*.db" -size +20M -delete
tail -c 1500 $O/bench_final.json

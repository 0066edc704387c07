# Round-3 profiling pass (run through gpurun from the repo root): rocprofv3 kernel statistics of the bench, the sweep kernel
# (+ the two HBM-traffic PMC passes and two SQ passes, each in its own run as MI355X_MICROARCH.md prescribes), the batch solvers,
# and one frame's dispatch timeline.  Summaries land in gpurun_out/r03/ (copied into profiles/ afterwards).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
timeout 250 rocprofv3 --kernel-trace --stats -d $O/prof_bench -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-host-inputs --no-batch > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof_sweep -- python $R/tools/sweep_only.py > $O/sweep.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -- python $R/tools/sweep_only.py > $O/pmc_fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -- python $R/tools/sweep_only.py > $O/pmc_write.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_sq1 -- python $R/tools/sweep_only.py > $O/pmc_sq1.log 2>&1
timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/pmc_sq2 -- python $R/tools/sweep_only.py > $O/pmc_sq2.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_ba_bench -- python $R/tools/ba_probe.py 60 30000 5 800 5 0 > $O/ba_bench.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_ba_large -- python $R/tools/ba_probe.py 200 600000 10 1500 3 0 > $O/ba_large.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_dense -- python $R/tools/dense_probe.py > $O/dense.log 2>&1
cd $R
for n in bench sweep ba_bench ba_large dense; do DB=$(find $O/prof_$n -name "*.db" | head -1); python tools/rocprof_summary.py $DB 40 > $O/${n}_kernel_stats.txt 2>&1; done
python tools/pmc_summary.py k_sweep_tile $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) > $O/sweep_pmc_hbm_traffic.txt 2>&1
python tools/pmc_summary.py k_finalize_pose $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) > $O/finalize_pmc_hbm_traffic.txt 2>&1
python tools/pmc_summary.py k_posepose $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) > $O/posepose_pmc_hbm_traffic.txt 2>&1
grep "^n_eb" $O/sweep.log | tail -1 >> $O/sweep_pmc_hbm_traffic.txt
python tools/pmc_counters.py k_sweep_tile $(find $O/pmc_sq1 -name "*.db" | head -1) $(find $O/pmc_sq2 -name "*.db" | head -1) > $O/sweep_sq_counters.txt 2>&1
DB=$(find $O/prof_bench -name "*.db" | head -1); python tools/rocprof_timeline.py $DB 40 2400 > $O/frame_timeline.txt 2>&1
cp $O/sweep_pmc_hbm_traffic.txt profiles/r03_sweep_pmc_hbm_traffic.txt      # (bench.py quotes the counter-based traffic of this very run)
timeout 400 python bench.py --replica-sweep 1,2,4,8 > $O/bench.json 2> $O/bench.err
find $O -name "*.db" -size +20M -delete
tail -c 400 $O/bench.json

# Round-5 K18 passes on the roofline graph (2.2 M static landmarks: 13.3 M edges, ~510 MB per sweep launch, twice the Infinity Cache): kernel statistics,
# the two HBM-traffic PMC passes (each in its own run, MI355X_MICROARCH.md), the SQ passes, and the phase probe (-DSWEEP_PROF build).
# usage (gpurun): bash tools/profile_round5_sweep.sh [tag]      -> gpurun_out/r05<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05${1:-p}
N=2200000
mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_sweep -- python $R/tools/sweep_only.py $N > $O/sweep.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -- python $R/tools/sweep_only.py $N > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -- python $R/tools/sweep_only.py $N > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_sq1 -- python $R/tools/sweep_only.py $N > $O/pmc_sq1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/pmc_sq2 -- python $R/tools/sweep_only.py $N > $O/pmc_sq2.log 2>&1
cd $R
DB=$(find $O/prof_sweep -name "*.db" | head -1); python tools/rocprof_summary.py $DB 40 > $O/sweep_kernel_stats.txt 2>&1
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
python tools/pmc_summary.py k_sweep_tile $F $W > $O/sweep_pmc_hbm_traffic.txt 2>&1
python tools/pmc_summary.py k_finalize_pose $F $W > $O/finalize_pmc_hbm_traffic.txt 2>&1
python tools/pmc_summary.py k_posepose $F $W > $O/posepose_pmc_hbm_traffic.txt 2>&1
grep "^n_eb" $O/sweep.log | tail -1 >> $O/sweep_pmc_hbm_traffic.txt
python tools/pmc_counters.py k_sweep_tile $(find $O/pmc_sq1 -name "*.db" | head -1) $(find $O/pmc_sq2 -name "*.db" | head -1) > $O/sweep_sq_counters.txt 2>&1
find $O -name "*.db" -size +20M -delete
VDO_HIP_LIB=$R/vdo_slam_amd/libvdo_hip_prof.so timeout 200 python tools/sweep_phase_probe.py $N > $O/sweep_phase_probe.txt 2>&1
cat $O/sweep_pmc_hbm_traffic.txt; head -12 $O/sweep_kernel_stats.txt; cat $O/sweep_sq_counters.txt | head -20; cat $O/sweep_phase_probe.txt

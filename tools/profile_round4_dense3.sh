# Round 4, dense solver, third pass: every launch sequence against the host Cholesky again, phase probes of k_chol_step and of the assembly kernel,
# the dense tests, ms per LM iteration.  -> gpurun_out/r04f/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 $R/tools/dense_check 5 > $O/dense_check.log 2>&1; echo "dense_check rc $?"; grep "2176\|BAD\|bad" $O/dense_check.log
timeout 100 $R/tools/dense_check_prof 3 4 > $O/dense_check_prof.log 2>&1; echo "dense_check_prof rc $?"; grep "prof\|2176" $O/dense_check_prof.log
cd $R
timeout 300 python -m pytest tests/test_ba_gpu.py tests/test_dense_check_gpu.py -m gpu -q --tb=short -rf -k "dense or non_path or wide_partial or tile_size" 2>&1 | grep -v "^  File \"/usr" | tail -15 > $O/test_ba.log; tail -4 $O/test_ba.log
VDO_HIP_LIB=$R/vdo_slam_amd/libvdo_hip_denseprof.so timeout 100 python tools/dense_asm_probe.py > $O/dense_asm_probe.log 2>&1; cat $O/dense_asm_probe.log | tail -12
DENSE_PROBE_VERSIONS=4 DENSE_PROBE_CHUNKS=2,4,8 timeout 200 python tools/dense_probe.py > $O/dense_probe.log 2>&1; tail -5 $O/dense_probe.log
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_probe -- env DENSE_PROBE_VERSIONS=4 python $R/tools/dense_probe.py > $O/dense_probe_prof.log 2>&1 )
DB=$(find $O/prof_probe -name "*.db" | head -1); python tools/rocprof_summary.py $DB 40 > $O/dense_probe_kernel_stats.txt 2>&1; cut -c1-150 $O/dense_probe_kernel_stats.txt | head -10
find $O -name "*.db" -size +20M -delete

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O
for v in "" _n1; do
  VDO_HIP_LIB=$R/vdo_slam_amd/libvdo_hip$v.so timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof$v -- python $R/tools/ba_probe.py 200 600000 10 1500 3 0 > $O/ba_large$v.log 2>&1
  DB=$(find $O/prof$v -name "*.db" | head -1); python $R/tools/rocprof_summary.py $DB 40 2>/dev/null | grep "pchain_factor\|pcg_chain<0>"
  grep -i "chi\|iter" $O/ba_large$v.log | tail -2
done
find $O -name "*.db" -delete

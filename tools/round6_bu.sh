#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r06bu; mkdir -p $O
cd $R
for v in 4_2 8_2 8_4 6_3 12_4; do
  fc=${v%_*}; bc=${v#*_}
  bash tools/build_variant.sh fc$v "-DVDO_FC=$fc -DVDO_BC=$bc" ba_solve > /dev/null 2>&1
  for i in 1 2; do
  VDO_HIP_LIB=$R/vdo_slam_amd/libvdo_hip_fc$v.so timeout 300 python tools/ba_variant_probe.py omd large 2>&1 | grep "ms/LM" | sed "s/^/fc$v /" | cut -c1-150 | tee -a $O/ab.txt
  done
done
for v in 4_2 8_4; do
  VDO_HIP_LIB=$R/vdo_slam_amd/libvdo_hip_fc$v.so timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr_$v -- python tools/ba_variant_probe.py omd > /dev/null 2>&1
  python tools/rocprof_summary.py $(find $O/tr_$v -name "*.db" | head -1) 30 2>/dev/null | grep "k_factor_chains" | sed "s/^/fc$v /" | cut -c1-120 | tee -a $O/ab.txt
  rm -rf $O/tr_$v
done

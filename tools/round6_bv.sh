#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r06bv; mkdir -p $O
cd $R
for pb in 128 64 32 16 8 4 0; do
  for i in 1 2; do
  VDO_BA_CHAINS_PER_BLOCK=$pb timeout 300 python tools/ba_variant_probe.py omd large bench 2>&1 | grep "ms/LM" | sed "s/^/per_block $pb: /" | cut -c1-150 | tee -a $O/ab.txt
  done
done
for pb in 128 0; do
  VDO_BA_CHAINS_PER_BLOCK=$pb timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr_$pb -- python tools/ba_variant_probe.py omd large > /dev/null 2>&1
  python tools/rocprof_summary.py $(find $O/tr_$pb -name "*.db" | head -1) 30 2>/dev/null | grep "k_factor_chains" | sed "s/^/per_block $pb: /" | cut -c1-120 | tee -a $O/ab.txt
  rm -rf $O/tr_$pb
done

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y; mkdir -p $O
timeout 900 python -m pytest tests/test_omd_gpu.py tests/test_ba_gpu.py -q -x 2>&1 | tail -3 | tee $O/tests.log
cd /tmp && export TMPDIR=/tmp
for m in twosided serial; do
  if [ $m = serial ]; then export VDO_BA_CHAIN_SERIAL=1; fi
  for tag in omd large; do
    if [ $tag = omd ]; then A="300 150000 4 40000 5 0"; else A="239 950000 20 500 3 0"; fi
    timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_$m -- python $R/tools/ba_probe.py $A > $O/probe_${tag}_$m.log 2>&1
    DB=$(find $O/prof_${tag}_$m -name "*.db" | head -1); python $R/tools/rocprof_summary.py $DB 40 > $O/${tag}_${m}_kernel_stats.txt 2>&1
    echo "== $tag $m"; grep -E "k_factor_chains|k_chain_inverse|k_sweep_tile<true" $O/${tag}_${m}_kernel_stats.txt | cut -c1-110; grep "LM its" $O/probe_${tag}_$m.log
  done
done
find $O -name "*.db" -delete

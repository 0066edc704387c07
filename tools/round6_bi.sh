#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r06bi; mkdir -p $O
cd $R
for c in 1 2 4 8; do
  export VDO_BA_DENSE_CHUNK=$c
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_$c -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-host-inputs --no-parity > $O/bench_$c.txt 2> $O/bench_$c.err
  python tools/rocprof_summary.py $(find $O/trace_$c -name "*.db" | head -1) 60 > $O/stats_$c.txt 2>&1
  echo "chunk $c: $(grep k_schur_dense_tile $O/stats_$c.txt | cut -c1-120)"
  rm -rf $O/trace_$c
done

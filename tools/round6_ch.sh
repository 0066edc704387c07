#!/bin/bash
cd /root/repo; O=gpurun_out/r06ch; mkdir -p $O
VDO_BATCH_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-host-inputs --no-parity > $O/b.json 2> $O/err.txt
grep "partial batch\|^\[batch\]" $O/err.txt | tail -9 | cut -c1-170

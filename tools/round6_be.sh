#!/bin/bash
cd /root/repo; O=gpurun_out/r06be; mkdir -p $O
VDO_BATCH_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-host-inputs --no-parity > $O/bench.json 2> $O/bench.err
grep "vdo_ba_create\|^\[batch\]\|partial batch" $O/bench.err | tail -9 | cut -c1-260

#!/bin/bash
O=gpurun_out/r06aq; mkdir -p $O
VDO_BATCH_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-batch --no-host-inputs --no-parity --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep "vdo_ba_create\]\|batch\]" $O/bench.err | sed -n 3,10p

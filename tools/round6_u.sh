#!/bin/bash
O=gpurun_out/r06u; mkdir -p $O
VDO_VERBOSE=2 VDO_BATCH_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-batch --no-host-inputs --no-parity --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep -n "trial\|iteration=\|batch\]" $O/bench.err | head -75

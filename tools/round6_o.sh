#!/bin/bash
O=gpurun_out/r06o; mkdir -p $O
VDO_PIPE_TRACE_OBJ=1 timeout 600 python bench.py --steps 60 --warmup 5 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-batch > $O/bench.json 2> $O/bench.err
grep "obj lm" $O/bench.err | head -150 > $O/obj_lm_trace.txt
python tools/step_events.py > $O/step_events.txt 2>&1
tail -5 $O/obj_lm_trace.txt

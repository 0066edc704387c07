#!/bin/bash
# round 6, first GPU call: parity of the bench sequence against the whole reference (probe), object-LM statistics of the bench sequence, GPU suite
O=gpurun_out/r06a; mkdir -p $O
( time timeout 900 python tools/bench_parity_probe.py 148 5 $O ) > $O/parity_probe.log 2>&1
tail -60 $O/parity_probe.log
( time timeout 300 python tools/bench_parity_probe.py 20 5 $O ) > $O/parity_probe20.log 2>&1
tail -45 $O/parity_probe20.log
VDO_PIPE_TRACE_OBJ=1 VDO_BENCH_DUMP_STEPS=1 timeout 600 python bench.py --no-batch --no-host-inputs --no-cpu-baseline > $O/bench_trace.json 2> $O/bench_trace.err
grep -c "obj lm" $O/bench_trace.err
bash tools/gpu_suite_by_file.sh $O/suite.log > $O/suite_summary.txt 2>&1
tail -70 $O/suite_summary.txt

#!/bin/bash
O=gpurun_out/r06i; mkdir -p $O
VDO_BENCH_SYNC_OBJECTS=1 VDO_CHAIN_TRACE=1 VDO_PNP_TRACE=1 timeout 600 python bench.py --steps 60 --warmup 5 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-batch > $O/bench_trace.json 2> $O/bench_trace.err
grep "vdo_object_chain:" $O/bench_trace.err | tail -3
grep "pnp trace" $O/bench_trace.err | tail -3
python -c "
import json
d=json.loads(open('$O/bench_trace.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['host_ms_per_section'], d['config']['step_ms_p50_p90_max'])"

#!/bin/bash
O=gpurun_out/r06l; mkdir -p $O
timeout 900 python -m pytest tests/test_flow2_gpu.py tests/test_lm_dev_gpu.py tests/test_golden_gpu.py tests/test_track_sequence_gpu.py tests/test_system_gpu.py -q -x 2>&1 | tail -6 | tee $O/tests.log
bash tools/build_profiled_flow2.sh > $O/build_prof.log 2>&1
VDO_HIP_LIB=$PWD/vdo_slam_amd/libvdo_hip_prof.so timeout 300 python tools/flow2_phase_probe.py 1200 o800 o400 o230 o120 2>&1 | tee $O/flow2_phase_probe.txt
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-batch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'deferred', d['value_deferred'], d['config']['step_ms_p50_p90_max'], d['config']['host_ms_per_section'])"; done | tee $O/bench_quick.txt

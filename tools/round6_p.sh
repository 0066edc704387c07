#!/bin/bash
O=gpurun_out/r06p; mkdir -p $O
run() { timeout 600 python bench.py --steps 20 --warmup 5 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-batch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 value', round(d['value'],1), 'deferred', round(d['value_deferred'],1), d['config']['step_ms_p50_p90_max'])"; }
run256() { timeout 600 python bench.py --steps 148 --warmup 5 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-batch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 148-step value', round(d['value'],1), 'deferred', round(d['value_deferred'],1), d['config']['step_ms_p50_p90_max'])"; }
for i in 1 2 3; do run t256; done | tee $O/bench_ab.txt
run256 t256 | tee -a $O/bench_ab.txt
touch vdo_slam_amd/csrc/flow2.hip; make -C vdo_slam_amd/csrc F2_THREADS=512 2>&1 | tail -2 > $O/build512.log; make -C vdo_slam_amd/host 2>&1 | tail -1 >> $O/build512.log
for i in 1 2 3; do run t512; done | tee -a $O/bench_ab.txt
run256 t512 | tee -a $O/bench_ab.txt
timeout 900 python -m pytest tests/test_flow2_gpu.py tests/test_lm_dev_gpu.py tests/test_golden_gpu.py tests/test_track_sequence_gpu.py tests/test_system_gpu.py tests/test_bench_sequence_gpu.py -q 2>&1 | tail -8 | tee $O/tests512.log
F2_THREADS=512 bash tools/build_profiled_flow2.sh > $O/build_prof.log 2>&1
VDO_HIP_LIB=$PWD/vdo_slam_amd/libvdo_hip_prof.so timeout 300 python tools/flow2_phase_probe.py 1200 o800 o400 o230 o120 2>&1 | grep -v amdgpu.ids | tee $O/flow2_phase_probe_512.txt

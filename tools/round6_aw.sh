#!/bin/bash
O=gpurun_out/r06aw; mkdir -p $O
run() { timeout 600 python bench.py --steps 20 --warmup 5 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-batch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 value', round(d['value'],1), 'deferred', round(d['value_deferred'],1), d['config']['step_ms_p50_p90_max'])"; }
for i in 1 2 3; do run xscope2; done | tee $O/xscope_ab.txt
touch vdo_slam_amd/csrc/flow2.hip; make -C vdo_slam_amd/csrc F2_THREADS="256 -DF2_XSCOPE=1" 2>&1 | tail -1 > $O/build.log; make -C vdo_slam_amd/host 2>&1 | tail -1 >> $O/build.log
for i in 1 2 3; do run xscope1; done | tee -a $O/xscope_ab.txt
timeout 900 python -m pytest tests/test_flow2_gpu.py tests/test_track_sequence_gpu.py tests/test_system_gpu.py tests/test_bench_sequence_gpu.py -q 2>&1 | tail -3 | tee $O/tests_xscope1.log
F2_THREADS=256 F2_EXTRA=-DF2_XSCOPE=1 bash tools/build_profiled_flow2.sh > $O/build_prof.log 2>&1
VDO_HIP_LIB=$PWD/vdo_slam_amd/libvdo_hip_prof.so timeout 300 python tools/flow2_phase_probe.py 1200 o400 2>&1 | grep -v amdgpu.ids | tee $O/probe_xscope1.txt

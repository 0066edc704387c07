#!/bin/bash
cd /root/repo; O=gpurun_out/r06ce; mkdir -p $O
VDO_BATCH_TRACE=1 timeout 600 python tools/ba_variant_probe.py bench omd large roof > $O/out.txt 2> $O/err.txt
grep "vdo_ba_create" $O/err.txt | cut -c1-220; cut -c1-200 $O/out.txt
timeout 1500 python -m pytest tests/test_ba_gpu.py tests/test_omd_gpu.py tests/test_windowed_ba_gpu.py tests/test_golden_gpu.py -x -q -m gpu > $O/t.log 2>&1; tail -3 $O/t.log

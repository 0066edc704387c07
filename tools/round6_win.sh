#!/bin/bash
O=gpurun_out/r06f; mkdir -p $O
timeout 1500 python -m pytest tests/test_windowed_ba_gpu.py tests/test_bench_sequence_gpu.py tests/test_track_sequence_gpu.py -m gpu -q -x -s 2>&1 | grep -v "^iteration=" | tail -30 | tee $O/tests.log

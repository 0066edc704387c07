#!/bin/bash
O=gpurun_out/r06ab; mkdir -p $O
( time timeout 2000 python -m pytest tests/test_windowed_ba_gpu.py -q -x -s 2>&1 | tail -25 ) 2>&1 | tee $O/tests.log | tail -30

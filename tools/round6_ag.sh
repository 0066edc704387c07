#!/bin/bash
O=gpurun_out/r06ag; mkdir -p $O
timeout 900 python -m pytest tests/test_ba_gpu.py -q -x -k "kitti_length" 2>&1 | tail -15 | tee $O/hub.log
timeout 1200 python -m pytest tests/test_ba_gpu.py tests/test_edge_cases_gpu.py tests/test_omd_gpu.py tests/test_dist.py -q -x 2>&1 | tail -4 | tee $O/tests.log

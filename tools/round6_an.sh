#!/bin/bash
O=gpurun_out/r06an; mkdir -p $O
timeout 900 python -m pytest tests/test_ba_gpu.py -q -x -k "more_than_128 or hub or kitti_length" 2>&1 | tail -15 | tee $O/long.log

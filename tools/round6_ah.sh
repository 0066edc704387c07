#!/bin/bash
O=gpurun_out/r06ah; mkdir -p $O
timeout 900 python -m pytest tests/test_ba_gpu.py -q -k "kitti_length or hub" 2>&1 | tail -5 | tee $O/hub.log

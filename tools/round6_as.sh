#!/bin/bash
O=gpurun_out/r06as; mkdir -p $O
timeout 600 python -m pytest tests/test_ba_gpu.py -q -x -k "every_tile_size" 2>&1 | grep -v "^  File" | tail -30 | cut -c1-400 | tee $O/fail.log

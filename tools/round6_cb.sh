#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r06cb
timeout 900 python -m pytest tests/test_dist.py -x -q -m gpu > gpurun_out/r06cb/t.log 2>&1; tail -5 gpurun_out/r06cb/t.log

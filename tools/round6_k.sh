#!/bin/bash
O=gpurun_out/r06k; mkdir -p $O
timeout 300 python -m pytest tests/test_ba_gpu.py -q -k "config4" 2>&1 | grep -E "AssertionError|assert |passed|failed|Hpp|Hll|Hpl|Hlp" | head -20 | tee $O/config4.log
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/suite.log

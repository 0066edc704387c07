#!/bin/bash
O=gpurun_out/r06aj; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -22 | tee $O/suite.log

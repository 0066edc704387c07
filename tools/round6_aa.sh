#!/bin/bash
O=gpurun_out/r06aa; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/suite.log

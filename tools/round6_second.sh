#!/bin/bash
O=gpurun_out/r06b; mkdir -p $O
timeout 600 python tools/bench_divergence_probe.py 20 5 own > $O/div_own.log 2>&1; tail -40 $O/div_own.log
timeout 600 python tools/bench_divergence_probe.py 20 5 product > $O/div_product.log 2>&1; tail -40 $O/div_product.log
timeout 900 python -m pytest tests/test_omd_gpu.py -m gpu -q -x -s 2>&1 | tail -15

#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_dense_check_gpu.py -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2

#!/bin/bash
O=gpurun_out/r06ai; mkdir -p $O
bash tools/gpu_suite_by_file.sh $O/suite.log > $O/suite_summary.txt 2>&1; grep -c passed $O/suite_summary.txt; grep "failed\|FAILED\|ERROR" $O/suite_summary.txt | head
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3

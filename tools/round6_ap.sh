#!/bin/bash
O=gpurun_out/r06ap; mkdir -p $O
bash tools/gpu_suite_by_file.sh $O/suite.log > $O/suite_summary.txt 2>&1; grep -c "passed" $O/suite_summary.txt; grep "failed\|FAILED\|ERROR" $O/suite_summary.txt | head -5

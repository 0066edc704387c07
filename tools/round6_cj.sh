#!/bin/bash
cd /root/repo; O=gpurun_out/r06cj; mkdir -p $O
for i in 1 2; do ( time timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/suite_$i.log 2>&1 ) 2> $O/time_$i.txt; tail -2 $O/suite_$i.log; grep real $O/time_$i.txt; done

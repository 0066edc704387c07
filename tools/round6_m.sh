#!/bin/bash
O=gpurun_out/r06m; mkdir -p $O
for t in default 0 1 7; do if [ $t = default ]; then python tools/orb_probe.py; else VDO_ORB_THREADS=$t python tools/orb_probe.py; fi; done 2>&1 | grep -v amdgpu.ids | tee $O/orb_probe.txt

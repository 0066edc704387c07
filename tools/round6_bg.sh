#!/bin/bash
cd /root/repo; O=gpurun_out/r06bg; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/dense_check.hip -o /tmp/dense_check 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DDENSE_PROF tools/dense_check.hip -o /tmp/dense_check_prof 2>/dev/null
timeout 120 /tmp/dense_check 1 0 small > $O/small.txt 2>&1; cat $O/small.txt
timeout 120 /tmp/dense_check_prof 1 0 small > $O/small_prof.txt 2>&1; grep -A1 "n0=120 spd\|n0=36" $O/small_prof.txt

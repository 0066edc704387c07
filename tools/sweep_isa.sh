#!/bin/bash
# Debug aid: ISA of k_sweep_tile<true, true> -> /tmp/sweep_true.s, its register use, and the order of its memory instructions / waits in the head of a tile.
cd "$(dirname "$0")/../vdo_slam_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -Wno-unused-value $1 -S --cuda-device-only -o /tmp/sweep.s ba_sweep.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "k_sweep_tileILb1ELb1" | grep "VGPRs:\|Spill\|Occupancy\|error"
awk '/^_ZN3vdo12k_sweep_tileILb1ELb1EEEvNS_5BADevEi:/,/s_endpgm/' /tmp/sweep.s > /tmp/sweep_true.s
wc -l /tmp/sweep_true.s
grep -n "global_load\|flat_load\|s_waitcnt vmcnt\|s_barrier\|s_cbranch\|ds_write" /tmp/sweep_true.s | head -${2:-60}

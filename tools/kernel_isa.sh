#!/bin/bash
# Debug aid: tools/kernel_isa.sh SOURCE(without .hip) MANGLED_NAME_PATTERN [lines] [extra flags] -> /tmp/kernel.s = ISA of the first kernel whose
# mangled name matches; prints its register use and the order of its memory instructions, waits, barriers and branches.
cd "$(dirname "$0")/../vdo_slam_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -ffp-contract=off -Wno-unused-value $4 -S --cuda-device-only -o /tmp/all.s $1.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "Function Name: .*$2" | grep "Function Name\|VGPRs:\|Spill\|Occupancy\|error" | head -5
N=$(grep -m1 "^_Z[A-Za-z0-9_]*$2[A-Za-z0-9_]*:" /tmp/all.s | sed 's/:.*//')
awk -v n="$N:" 'index($0, n)==1,/s_endpgm/' /tmp/all.s > /tmp/kernel.s
echo "$N: $(wc -l < /tmp/kernel.s) lines"
grep -n "global_load\|flat_load\|s_waitcnt vmcnt\|s_barrier\|s_cbranch\|ds_write\|global_store" /tmp/kernel.s | head -${3:-60}

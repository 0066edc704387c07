#!/bin/bash
cd /root/repo; O=gpurun_out/r06bj; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/sync_latency_probe.hip -o /tmp/sync_probe 2>/dev/null
timeout 120 /tmp/sync_probe | tee $O/sync_probe.txt

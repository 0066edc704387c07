#!/bin/bash
cd /root/repo; O=gpurun_out/r06bz; mkdir -p $O
for tol in 1e-10 1e-9 1e-8 1e-7 1e-6; do
  VDO_BA_PCG_TOL=$tol timeout 300 python tools/ba_variant_probe.py omd large bench config3 2>&1 | grep "ms/LM" | sed "s/^/tol $tol: /" | cut -c1-220 | tee -a $O/ab.txt
done

#!/bin/bash
cd /root/repo; O=gpurun_out/r06cf; mkdir -p $O
VDO_BATCH_TRACE=1 timeout 600 python tools/ba_variant_probe.py bench large > $O/out.txt 2> $O/err.txt
grep "vdo_ba_create" $O/err.txt | cut -c1-260

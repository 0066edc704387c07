#!/bin/bash
cd /root/repo; O=gpurun_out/r06cd; mkdir -p $O
for m in 1 2 0; do
  VDO_BA_PLACE=$m VDO_BATCH_TRACE=1 timeout 600 python tools/ba_variant_probe.py bench large roof > $O/out_$m.txt 2> $O/err_$m.txt
  echo "== VDO_BA_PLACE=$m"; grep "vdo_ba_create" $O/err_$m.txt | sed 's/.*tiles built in/tiles built in/' | cut -c1-100; cut -c1-160 $O/out_$m.txt
done

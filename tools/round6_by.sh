#!/bin/bash
cd /root/repo; O=gpurun_out/r06by; mkdir -p $O
for i in 1 2 3; do
for v in fresh shared; do
  if [ $v = shared ]; then export VDO_ORB_ON_LM_STREAM=1; else unset VDO_ORB_ON_LM_STREAM; fi
  timeout 300 python tools/step_events.py 20 > $O/ev_$v.txt 2>/dev/null
  echo "$v: $(grep 'defer_objects' $O/ev_$v.txt | tr '\n' ' ')" | tee -a $O/ab.txt
done; done
sed -n '/defer_objects=0/,/sections/p' $O/ev_shared.txt | tail -18

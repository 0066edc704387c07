#!/bin/bash
cd /root/repo; O=gpurun_out/r06bq; mkdir -p $O
run() {  # name queues prio
  for i in 1 2; do
  GPU_MAX_HW_QUEUES=$2 VDO_CTX_PRIO=$3 timeout 300 python tools/step_events.py 20 > $O/ev_$1.txt 2>/dev/null
  echo "$1 (queues $2 prio $3): $(grep 'defer_objects=0' $O/ev_$1.txt | tr '\n' ' ')" | tee -a $O/ab.txt
  done
  sed -n '/defer_objects=0/,/sections/p' $O/ev_$1.txt | head -17 | grep "obj_chain\|obj_lm_launched\|obj_lm_fetched\|orb_done\|cam_stage_done\|step_end" | tr '\n' ';'; echo
}
run base4 4 0,0,0,0,0
run base8 8 0,0,0,0,0
run lmhigh8 8 0,-1,-1,0,0
run lmhigh_orblow8 8 0,-1,-1,0,1
run lmhigh4 4 0,-1,-1,0,0
run lmhigh_orblow4 4 0,-1,-1,0,1
run mainlmhigh_orblow8 8 -1,-1,-1,0,1

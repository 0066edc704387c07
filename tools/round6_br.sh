#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r06br; mkdir -p $O
cd $R
run() {
  GPU_MAX_HW_QUEUES=$2 VDO_CTX_PRIO=$3 timeout 300 rocprofv3 --kernel-trace -d $O/tr_$1 -- python tools/step_events.py 20 > $O/ev_$1.txt 2>/dev/null
  echo "== $1 (queues $2 prio $3): $(grep 'defer_objects=0' $O/ev_$1.txt | tr '\n' ' ')"
  python tools/queue_map.py $O/tr_$1 2>&1 | cut -c1-400
  rm -rf $O/tr_$1
}
run base4 4 0,0,0,0,0
run base8 8 0,0,0,0,0
run lmhigh_orblow4 4 0,-1,-1,0,1

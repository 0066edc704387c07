#!/bin/bash
cd /root/repo; O=gpurun_out/r06bn; mkdir -p $O
VDO_CHAIN_TRACE=1 timeout 600 python tools/step_events.py 60 > $O/events.txt 2> $O/chain_trace.txt; grep "vdo_object_chain" $O/chain_trace.txt | tail -6; head -20 $O/events.txt

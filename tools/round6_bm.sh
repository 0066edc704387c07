#!/bin/bash
cd /root/repo; O=gpurun_out/r06bm; mkdir -p $O
timeout 600 python tools/step_events.py 20 > $O/events20.txt 2>&1; tail -50 $O/events20.txt

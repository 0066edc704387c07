#!/bin/bash
# round 5, GPU: after the EPnP change - the GPU suite file by file, then the per-frame legs of the bench only at 20 and at 148 steps
# usage (gpurun): bash tools/round5_epnp_check.sh [tag]
O=gpurun_out/r05${1:-g}; mkdir -p $O
bash tools/gpu_suite_by_file.sh $O/suite.log > $O/suite_summary.txt 2>&1
tail -60 $O/suite_summary.txt
for st in 20 148; do
  VDO_BENCH_NO_SHARDED=1 python bench.py --steps $st --no-batch > $O/bench_$st.json 2> $O/bench_$st.err || tail -5 $O/bench_$st.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$st.json").read().strip().splitlines()[-1]); c=d["cpu_baseline"]
print("steps", d["steps"], "value", d["value"], "x cpu", d["value"]/c["value"], "deferred", d.get("value_deferred"), "host_sync", d.get("value_host_inputs_sync"), "cpu", c["value"])
print(d["config"].get("per_frame_mean"), d["config"].get("object_motion_error_m_last_frame"), d["config"].get("trajectory_drift_m"))
PY
done

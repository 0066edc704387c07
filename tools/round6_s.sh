#!/bin/bash
O=gpurun_out/r06s; mkdir -p $O
timeout 900 python -m pytest tests/test_dist.py tests/test_ba_gpu.py tests/test_omd_gpu.py -q -x 2>&1 | tail -4 | tee $O/tests.log
for m in merged unmerged; do
if [ $m = unmerged ]; then export VDO_BA_NO_EXCHANGE_MERGE=1; fi
timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-live-pmc > $O/bench_gpus2_$m.json 2> $O/bench_gpus2_$m.err; tail -2 $O/bench_gpus2_$m.err | cut -c1-300
python - <<PY
import json
d=json.loads(open("$O/bench_gpus2_$m.json").read().strip().splitlines()[-1])
for k,v in d.get("sharded",{}).items(): print("$m", k, {q:v.get(q) for q in ("ms_per_lm_iter_sharded","ms_per_lm_iter_1gpu","same_trajectory_as_1gpu","allreduces_per_lm_iter","allreduce_bytes_per_lm_iter","lm_iterations","trials","error")})
PY
done | tee $O/sharded_ab.txt

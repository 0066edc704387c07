#!/bin/bash
O=gpurun_out/r06j; mkdir -p $O
timeout 900 python -m pytest tests/test_dist.py tests/test_ba_gpu.py tests/test_edge_cases_gpu.py -q -x 2>&1 | tail -6 | tee $O/tests.log
timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-live-pmc > $O/bench_gpus2.json 2> $O/bench_gpus2.err; tail -3 $O/bench_gpus2.err
python - <<PY
import json
d=json.loads(open("$O/bench_gpus2.json").read().strip().splitlines()[-1])
for k,v in d.get("sharded",{}).items(): print(k, {q:v.get(q) for q in ("ms_per_lm_iter_sharded","ms_per_lm_iter_1gpu","same_trajectory_as_1gpu","allreduces_per_lm_iter","allreduce_bytes_per_lm_iter","lm_iterations","trials","error")})
PY
for t in 3 0; do VDO_PNP_THREADS=$t timeout 600 python bench.py --steps 20 --warmup 5 --no-parity --no-full-sequence --no-host-inputs --no-cpu-baseline --no-batch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('VDO_PNP_THREADS=$t value', d['value'], 'deferred', d['value_deferred'], d['config']['host_ms_per_section']['ransac_obj'])"; done | tee $O/pnp_threads_ab.txt

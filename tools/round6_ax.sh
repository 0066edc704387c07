#!/bin/bash
O=gpurun_out/r06ax; mkdir -p $O
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_torchrun2.json 2> $O/bench_torchrun2.err ) 2> $O/time.txt; tail -3 $O/time.txt; tail -2 $O/bench_torchrun2.err | cut -c1-300
python - <<PY
import json
lines=[l for l in open("$O/bench_torchrun2.json").read().strip().splitlines() if l.startswith("{")]
print("json lines on stdout:", len(lines))
d=json.loads(lines[-1]); print(d["metric"][:40], d["value"], d["n_gpus"], d["steps"], d["scaling"], list(d.get("sharded",{}).keys()), {k:v.get("allreduces_per_lm_iter") for k,v in d.get("sharded",{}).items()})
PY

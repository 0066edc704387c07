#!/bin/bash
cd /root/repo; O=gpurun_out/r06cg; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/time.txt; tail -3 $O/time.txt
python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print(d["value"], d["n_gpus"], d["steps"], d["warmup"], d["ms_per_step"], d["scaling"], d["vs_baseline"], d["dtype"][:20], d["roofline"]["frac"], d["cpu_baseline"]["value"], d.get("value_with_windowed_ba"), {k:round(v,3) for k,v in d.items() if k.startswith("ms_per_lm")})
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1

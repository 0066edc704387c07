#!/bin/bash
O=gpurun_out/r06ac; mkdir -p $O
for m in auto densefirst; do
if [ $m = densefirst ]; then export VDO_BA_TINY_DENSE_FIRST=1; fi
VDO_BATCH_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-batch --no-host-inputs --no-parity --no-cpu-baseline > $O/bench_$m.json 2> $O/bench_$m.err
echo "== $m"; grep "batch\]" $O/bench_$m.err | head -6
python - <<PY
import json
d=json.loads(open("$O/bench_$m.json").read().strip().splitlines()[-1]); print("value_with_windowed_ba", d.get("value_with_windowed_ba"))
PY
done

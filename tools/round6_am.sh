#!/bin/bash
O=gpurun_out/r06am; mkdir -p $O
for m in spec nospec spec nospec; do
if [ $m = nospec ]; then export VDO_BA_NO_SPEC_LIN=1; else unset VDO_BA_NO_SPEC_LIN; fi
VDO_BATCH_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-batch --no-host-inputs --no-parity --no-cpu-baseline > $O/bench_$m.json 2> $O/bench_$m.err
echo "== $m"; grep "batch\]" $O/bench_$m.err | sed -n 2,5p
python - <<PY
import json
d=json.loads(open("$O/bench_$m.json").read().strip().splitlines()[-1]); print("value_with_windowed_ba", d.get("value_with_windowed_ba"), "full", d.get("value_full_sequence"))
PY
done

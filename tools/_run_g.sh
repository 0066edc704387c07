cd $GRAFT_REPO_ROOT
python -m pytest tests/test_track_sequence_gpu.py tests/test_system_gpu.py tests/test_host_classes_gpu.py -x -q -m gpu 2>&1 | tail -8
python -m pytest tests/test_ba_gpu.py -x -q -m gpu -k "partitioned" 2>&1 | tail -4
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-batch --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: round(d[k],1) for k in ('value','value_sync','value_host_inputs','value_host_inputs_sync') if k in d})
"; done
VDO_PIPE_NO_CAM_AHEAD=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-batch --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('no-ahead', {k: round(d[k],1) for k in ('value','value_sync','value_host_inputs','value_host_inputs_sync') if k in d})
"
python bench.py --gpus 1 --steps 60 --warmup 5 --no-batch --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('60 steps', {k: round(d[k],1) for k in ('value','value_sync','value_host_inputs','value_host_inputs_sync') if k in d})
"

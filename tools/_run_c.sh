cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --no-batch --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(d[k],1) for k in ('value','value_sync','value_host_inputs','value_host_inputs_sync')})"

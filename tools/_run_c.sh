cd $GRAFT_REPO_ROOT
VDO_PNP_TRACE=1 python bench.py --no-batch --no-cpu-baseline --no-host-inputs --steps 60 2>&1 >/dev/null | grep "pnp trace"
VDO_PNP_TRACE=1 VDO_PNP_THREADS=0 python bench.py --no-batch --no-cpu-baseline --no-host-inputs --steps 60 2> gpurun_out/t.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no pnp threads:', d['value'], d['config']['host_ms_per_section']['ransac_obj'])"; grep "pnp trace" gpurun_out/t.err

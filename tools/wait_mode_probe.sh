# Per-frame bench (GPU legs only) under the HIP runtime's wait / queue switches: does hipStreamSynchronize's wake-up latency sit on the object chain?
# -> gpurun_out/r04h/   (run through gpurun from the repo root)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h; mkdir -p $O
cd $R
run() { name=$1; shift; env "$@" timeout 200 python bench.py --no-cpu-baseline --no-batch --no-host-inputs > $O/bench_$name.json 2> $O/bench_$name.err; python - "$name" "$O/bench_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value", round(d["value"], 1), "deferred", round(d.get("value_deferred", 0), 1), "ms_per_step", round(d["ms_per_step"], 4))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
run base0 VDO_DUMMY=1
run wait1000 ROC_ACTIVE_WAIT_TIMEOUT=1000
run wait1000_q8 ROC_ACTIVE_WAIT_TIMEOUT=1000 GPU_MAX_HW_QUEUES=8
run wait1000_kernarg ROC_ACTIVE_WAIT_TIMEOUT=1000 HIP_FORCE_DEV_KERNARG=1
run wait100000 ROC_ACTIVE_WAIT_TIMEOUT=100000
run base1 VDO_DUMMY=1

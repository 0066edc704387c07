cd $GRAFT_REPO_ROOT
VDO_BENCH_DUMP_STEPS=1 VDO_PIPE_TRACE_SLOW=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-batch --no-cpu-baseline 2>&1 >/dev/null | grep "host-input\|slow step" | cut -c1-400

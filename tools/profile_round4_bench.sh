R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O
cd $R
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-300 $O/bench.json

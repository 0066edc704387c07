# Round 4, closing pass: the GPU suite file by file and the default bench run (-> gpurun_out/r04g/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O
cd $R
bash tools/gpu_suite_by_file.sh $O/suite.log 2>&1 | tail -40
timeout 700 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-400 $O/bench.json

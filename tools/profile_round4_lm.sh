# Round 4: the batch tests after the LM's reuse of the accepted trial's errors, ms per LM iteration on the bench's three graph shapes  -> gpurun_out/r04j/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_ba_gpu.py tests/test_g2o_replay_gpu.py tests/test_dist.py tests/test_host_classes_gpu.py -m gpu -q --tb=short -rf -x 2>&1 | grep -v "^  File \"/usr" | tail -12 > $O/tests.log; tail -5 $O/tests.log
for n in bench large; do timeout 150 python tools/ba_variant_probe.py $n > $O/ba_$n.log 2>&1; tail -2 $O/ba_$n.log; done
for n in bench large; do VDO_BA_LM_RECHECK=1 timeout 150 python tools/ba_variant_probe.py $n > $O/ba_${n}_recheck.log 2>&1; tail -1 $O/ba_${n}_recheck.log; done

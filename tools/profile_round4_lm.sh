# Round 4: the batch tests after the LM's reuse of the accepted trial's errors  -> gpurun_out/r04j/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_ba_gpu.py tests/test_g2o_replay_gpu.py tests/test_dist.py tests/test_host_classes_gpu.py tests/test_track_to_batch_gpu.py -m gpu -q --tb=short -rf -k "not dense and not non_path and not wide_partial and not tile_size and not config4" 2>&1 | grep -v "^  File \"/usr" | tail -12 > $O/tests2.log; tail -5 $O/tests2.log

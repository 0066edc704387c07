cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04k
timeout 120 python -m pytest tests/test_system_gpu.py tests/test_track_sequence_gpu.py tests/test_pipeline_gpu.py tests/test_edge_cases_gpu.py -m gpu -q --tb=short -rf 2>&1 | grep -v "^  File \"/usr" | tail -8 | tee gpurun_out/r04k/tests.log

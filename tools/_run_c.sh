cd $GRAFT_REPO_ROOT
python -m pytest tests/test_system_gpu.py tests/test_host_classes_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -5

# GPU check of the batch path (run through gpurun from the repo root): the BA / replay / host-class / distributed GPU tests, then the batch legs of bench.py
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ba_gpu.py tests/test_g2o_replay_gpu.py tests/test_host_classes_gpu.py tests/test_track_to_batch_gpu.py tests/test_dist.py -x -q -m gpu 2>&1 | tail -8
python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-host-inputs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in d:
    if 'lm_iter' in k or 'pcg' in k: print(k, d[k])
for k in d['config']:
    if 'lm_iter' in k or 'pcg' in k or 'batch' in k: print(k, d['config'][k])
"

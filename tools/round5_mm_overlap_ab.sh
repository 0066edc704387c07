#!/bin/bash
# round 5, GPU: the motion-model inlier count under the RANSAC kernels - sequence tests, then A/B of the per-frame legs on one box
timeout 900 python -m pytest tests/test_ransac_gpu.py tests/test_pipeline_gpu.py tests/test_track_sequence_gpu.py tests/test_system_gpu.py tests/test_capi_symbols.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do for combo in "" "VDO_PIPE_NO_MM_OVERLAP=1"; do
  env $combo python bench.py --steps 60 --no-batch --no-cpu-baseline --no-host-inputs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$combo]', 'value %.1f deferred %.1f' % (d['value'], d['value_deferred']), d['config'].get('step_ms_p50_p90_max'))"
done; done | tee gpurun_out/r05_mm_overlap_ab.txt

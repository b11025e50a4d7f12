#!/bin/bash
# bench value against the number of GPU-stage threads (frames in flight on the GPU) of the pipeline
for g in ${@:-6 8 10 12}; do
  for r in 1 2; do
    PCC_PIPELINE_GPU_THREADS=$g python bench.py --no-cpu-baseline --no-host-input 2>/dev/null |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gpu threads', $g, ' value', d['value'], ' gpu only', d['gpu_only_mpoints_per_s'], d['host_cpu_ms_per_frame'])"
  done
done

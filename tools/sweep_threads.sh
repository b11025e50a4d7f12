#!/bin/bash
# Sweep GPU-stage threads x entropy workers of the frame pipeline (bench.py, no CPU baseline).  Run on the GPU box.
for g in ${GT:-6 8 10 12}; do
  for w in ${WK:-12 14 16}; do
    v=$(PCC_PIPELINE_GPU_THREADS=$g python bench.py --steps ${STEPS:-1024} --warmup 16 --workers $w --no-cpu-baseline 2>/dev/null | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_ms_per_frame']; print('%.0f  launch %.2f finish %.2f entropy %.2f' % (d['value'], h['launch_call'], h['finish_call'], h['entropy_call']))")
    echo "gpu_threads=$g workers=$w: $v"
  done
done

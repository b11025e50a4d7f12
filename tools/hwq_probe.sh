#!/bin/bash
# saturated GPU stage against the number of hardware queues the HIP runtime spreads its streams over (GPU_MAX_HW_QUEUES, default 4)
OUT=gpurun_out/${1:-r02_hwq}; mkdir -p $OUT
for q in 1 2 3 4; do
  echo "== GPU_MAX_HW_QUEUES=$q" | tee -a $OUT/hwq.txt
  GPU_MAX_HW_QUEUES=$q python tools/gpu_throughput.py cfg2 4 8 10 12 16 2>&1 | tee -a $OUT/hwq.txt
done

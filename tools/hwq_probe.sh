#!/bin/bash
# saturated GPU stage against the number of hardware queues the HIP runtime spreads its streams over (GPU_MAX_HW_QUEUES, default 4)
OUT=gpurun_out/${1:-r02_hwq}; mkdir -p $OUT
for q in 5 6 4; do
  echo "== GPU_MAX_HW_QUEUES=$q" | tee -a $OUT/hwq.txt
  GPU_MAX_HW_QUEUES=$q python tools/gpu_throughput.py cfg2 10 12 15 16 18 2>&1 | tee -a $OUT/hwq.txt
done

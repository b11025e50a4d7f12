#!/bin/bash
OUT=gpurun_out/${1:-r02_w8}; mkdir -p $OUT
for w in 1 0 1 0; do
  export PCC_BENCH_SHAPE_WARMUP=$w
  python bench.py --no-cpu-baseline --no-host-input 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shape warm-up $w: value', d['value'], 'gpu_only', d['gpu_only_mpoints_per_s'], 'cpu', d['host_cpu_ms_per_frame'], 'thr', d['cgroup_throttled_ms_in_timed_region'])" | tee -a $OUT/ab.txt
  for k in 1 2 3 4 5; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-input 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   20 steps: value', d['value'], 'finish cpu', d['host_cpu_ms_per_frame']['finish_call'], 'thr', d['cgroup_throttled_ms_in_timed_region'])" | tee -a $OUT/ab.txt; done
done

#!/bin/bash
# the driver's bench command, a few times: value, ms per step, CPU time of pcc_hotpath_finish per frame, and how long the
# job's cgroup was throttled (CPU quota) inside the timed region
# usage: bash tools/bench20.sh [runs]
for i in $(seq 1 ${1:-5}); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-input 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], ' ms/step', d['ms_per_step'], ' finish cpu ms', d['host_cpu_ms_per_frame']['finish_call'], ' throttled ms', d.get('cgroup_throttled_ms_in_timed_region'))"
done

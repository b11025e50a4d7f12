for w in 16 20 24 16 20 24; do
  for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --workers $w --no-cpu-baseline --no-host-input 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('workers $w value', d['value'], ' ms/step', d['ms_per_step'], ' throttled ms', d.get('cgroup_throttled_ms_in_timed_region'))"
  done
done
for w in 16 24; do python bench.py --workers $w --no-cpu-baseline --no-host-input 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1024 steps workers $w value', d['value'], ' throttled ms', d.get('cgroup_throttled_ms_in_timed_region'), 'gpu_only', d['gpu_only_mpoints_per_s'])"; done

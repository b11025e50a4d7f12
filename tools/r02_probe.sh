#!/bin/bash
# quick probe: parity subset, single-stream latency (cfg2, cfg4), saturated throughput, kernel durations under load
TAG=${1:-probe}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_codec_golden.py tests/test_delta_gpu.py -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
python tools/gpu_latency.py cfg2 20 > $OUT/latency_cfg2.txt 2>&1; grep -v "^   (" $OUT/latency_cfg2.txt | tail -9
python tools/gpu_latency.py cfg4 8 > $OUT/latency_cfg4.txt 2>&1; grep -v "^   (" $OUT/latency_cfg4.txt | tail -8
python tools/queue_balance.py 2>&1 | tee $OUT/queue_balance.txt | grep "balanced"
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-input --steps 256 > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err)
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/bench_kernel_stats.txt && cat $OUT/bench_kernel_stats.txt
find $OUT -name '*.db' -size +20M -delete

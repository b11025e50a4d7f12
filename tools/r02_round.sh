#!/bin/bash
# One GPU-box session of round 2: parity tests, smoke, bench lines (driver's step count, default, cfg4),
# single-stream kernel traces and HBM-traffic counters for cfg2 and cfg4.
# Usage (through gpurun): bash tools/r02_round.sh <tag> [quick]
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as G; G.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "bench(20) rc=$?"; cat $OUT/bench_steps20.json
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
python bench.py --workload cfg4 --steps 48 --warmup 4 --no-host-input > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; echo "bench(cfg4) rc=$?"; cat $OUT/bench_cfg4.json
[ "$2" == "quick" ] && exit 0
for WL in cfg2 cfg4; do
  bash tools/prof_latency.sh $TAG/lat_$WL $WL > $OUT/latency_$WL.txt 2>&1; tail -16 $OUT/latency_$WL.txt
  bash tools/pmc_run.sh $TAG/pmc_$WL $WL > $OUT/pmc_$WL.log 2>&1; tail -3 $OUT/pmc_$WL.log
done
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-input --steps 256 > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err)
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/bench_kernel_stats.txt && cat $OUT/bench_kernel_stats.txt
find $OUT -name '*.db' -size +20M -delete

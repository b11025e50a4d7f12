#!/bin/bash
# One GPU-box session: parity tests, smoke, bench line, rocprofv3 kernel trace of the bench command.
# Usage (through gpurun): bash tools/gpu_round.sh <tag> [bench args...]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as G; G.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err)
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats.txt && cat $OUT/kernel_stats.txt
find $OUT/prof -name '*.db' -size +20M -delete
# single-stream latency + its kernel trace, and the HBM-traffic counters (separate --pmc passes)
bash tools/prof_latency.sh $TAG/lat > $OUT/latency_stats.txt 2>&1; tail -14 $OUT/latency_stats.txt
bash tools/pmc_run.sh $TAG/pmc > $OUT/pmc.log 2>&1; tail -3 $OUT/pmc.log

#!/bin/bash
# rocprofv3 kernel trace of the single-context latency probe -> per-kernel stats
TAG=${1:-lat}; WL=${2:-cfg2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/tools/gpu_latency.py $WL 20 > $OUT/latency.txt 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name '*.db' | head -1)
cat $OUT/latency.txt | head -3
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats.txt && cat $OUT/kernel_stats.txt
find $OUT/prof -name '*.db' -size +20M -delete

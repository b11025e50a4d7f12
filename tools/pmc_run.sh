#!/bin/bash
# HBM traffic per kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE do not fit one pass) over
# the single-stream latency probe.  Usage (through gpurun): bash tools/pmc_run.sh <tag> [workload]
# PMC_DISTINCT=12: that many distinct frames take turns (384 MB of cfg2 clouds: more than the Infinity Cache holds, so the
# figure is HBM traffic; with one frame the cloud is re-read from the cache)
TAG=${1:-pmc}; WL=${2:-cfg2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o pmc -- python $GRAFT_REPO_ROOT/tools/gpu_latency.py $WL ${PMC_FRAMES:-$(( ${PMC_DISTINCT:-1} > 6 ? 2 * ${PMC_DISTINCT:-1} : 6 ))} ${PMC_DISTINCT:-1} > $OUT/$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
find $OUT -name '*.csv' | head; python tools/pmc_traffic.py $OUT $WL

#!/bin/bash
# A/B of one environment knob -- or of two builds of the library: VAR=PCC_LIB, values = paths -- : the parity subset with
# EVERY setting (a rebuilt library is a different program), then lone-frame latency and the saturated GPU stage for both
# settings, twice; both timing tools hold every frame they time against the oracle's golden digests.
#   bash tools/ab_probe.sh <tag> <VAR> <value-A> <value-B> [workload]
TAG=$1; VAR=$2; A=$3; B=$4; WL=${5:-cfg2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for v in "$A" "$B"; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  L=$OUT/pytest_$(basename "$v").log
  python -m pytest tests/test_gpu_parity.py tests/test_codec_golden.py -m gpu -x -q > $L 2>&1; echo "$VAR=$v pytest rc=$?" | tee -a $L
  tail -2 $L
done
for v in "$A" "$B" "$A" "$B"; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  python tools/gpu_latency.py $WL 30 > $OUT/latency_${WL}_$(basename "$v").txt 2>&1
  echo "== $VAR=$v"; grep -A9 "profiling=2" $OUT/latency_${WL}_$(basename "$v").txt | grep -v "^   (" | head -10
  grep "profiling=0" $OUT/latency_${WL}_$(basename "$v").txt
  python tools/gpu_throughput.py $WL 12 2>&1 | tee -a $OUT/thr_${WL}_$(basename "$v").txt | tail -3
done

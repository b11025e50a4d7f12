#!/bin/bash
# A/B of two settings on one box -- each a list of KEY=value pairs for the environment, e.g. two builds of the library
# ("PCC_LIB=.../libpcc_hip.so" against "PCC_LIB=.../libpcc_hip_r02.so") or one developer build with and without a switch --:
# the parity subset with EVERY setting (another library is another program), then lone-frame latency and the saturated GPU
# stage for both settings, alternating, twice; both timing tools hold every frame they time against the oracle's golden digests.
#   bash tools/ab_probe.sh <tag> "<setting A>" "<setting B>" [workload]
TAG=$1; A=$2; B=$3; WL=${4:-cfg2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
name() { echo "$1" | sed 's|[^ ]*/||g; s/[ =]/_/g'; }
for v in "$A" "$B"; do
  L=$OUT/pytest_$(name "$v").log
  env $v python -m pytest tests/test_gpu_parity.py tests/test_codec_golden.py -m gpu -q --timeout 900 > $L 2>&1; echo "[$v] pytest rc=$?" | tee -a $L
  tail -2 $L
done
for v in "$A" "$B" "$A" "$B"; do
  N=$(name "$v")
  env $v python tools/gpu_latency.py $WL 30 > $OUT/latency_${WL}_$N.txt 2>&1
  echo "== [$v]"; grep -A9 "profiling=2" $OUT/latency_${WL}_$N.txt | grep -v "^   (" | head -10
  grep "profiling=0" $OUT/latency_${WL}_$N.txt
  env $v python tools/gpu_throughput.py $WL 12 2>&1 | tee -a $OUT/thr_${WL}_$N.txt | tail -3
done

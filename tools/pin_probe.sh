#!/bin/bash
# short calls of the pipeline: entropy threads unpinned / one core each / a group of cores each, and how the frames of a
# short call are spread over the threads (PCC_PIPELINE_SPREAD); medians over interleaved repetitions
# PCC_PIPELINE_SPREAD is a developer switch: only the developer build reads it (the shipped library would silently run its default)
OUT=gpurun_out/${1:-r02_pin}; mkdir -p $OUT; rm -f $OUT/calls.txt
export PCC_LIB=${PCC_LIB:-$PWD/cwi-pcl-codec_amd/libpcc_hip_dev.so} PCC_ALLOW_NON_PRODUCT_LIB=1
for round in 1 2 3; do
for v in none default; do
  unset PCC_PIPELINE_PIN PCC_PIPELINE_SPREAD
  case $v in none) export PCC_PIPELINE_PIN=none PCC_PIPELINE_SPREAD=2;; esac
  python tools/short_calls.py 20 20 20 64 256 1024 2>&1 | awk -v v=$v '{print v "_" $2, $4}' >> $OUT/calls.txt
done
done
python - $OUT/calls.txt <<'PY'
import sys, statistics as st
d={}
for l in open(sys.argv[1]):
    a=l.split(); d.setdefault(a[0],[]).append(float(a[1]))
for k,v in sorted(d.items()): print(k, "n=%d median %.2f mean %.2f min %.2f max %.2f ms"%(len(v), st.median(v), st.mean(v), min(v), max(v)))
PY

#!/bin/bash
# ktime stamps + rocprofv3 kernel trace of the single-stream latency probe (cfg2 and cfg4)
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
PCC_LIB=$PWD/cwi-pcl-codec_amd/libpcc_hip_ktime.so python tools/ktime.py > $OUT/ktime.txt 2>&1; cat $OUT/ktime.txt | grep -v "^pass\|stamps (max"
for WL in cfg2 cfg4; do
  bash tools/prof_latency.sh $TAG/lat_$WL $WL > $OUT/latency_$WL.txt 2>&1; cat $OUT/latency_$WL.txt | tail -14
done

#!/bin/bash
# Round-2 probe: GPU tests, single-stream latency for cfg2 / cfg4, in-kernel phase stamps (ktime build).
# Usage (through gpurun): bash tools/r02_probe.sh <tag> [notest]
TAG=${1:-probe}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" != "notest" ]; then
  python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -3 $OUT/pytest_gpu.log
fi
python tools/gpu_latency.py cfg2 30 > $OUT/lat_cfg2.txt 2>&1; cat $OUT/lat_cfg2.txt
python tools/gpu_latency.py cfg4 10 > $OUT/lat_cfg4.txt 2>&1; cat $OUT/lat_cfg4.txt
if [ -f cwi-pcl-codec_amd/libpcc_hip_ktime.so ]; then
  PCC_LIB=$PWD/cwi-pcl-codec_amd/libpcc_hip_ktime.so python tools/ktime.py > $OUT/ktime.txt 2>&1; cat $OUT/ktime.txt
fi

#!/bin/bash
# quick probe: parity subset, single-stream latency (cfg2, cfg4), saturated throughput
TAG=${1:-probe}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_codec_golden.py tests/test_delta_gpu.py -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
python tools/gpu_latency.py cfg2 20 > $OUT/latency_cfg2.txt 2>&1; grep -v "^   (" $OUT/latency_cfg2.txt | tail -14
python tools/gpu_latency.py cfg4 8 > $OUT/latency_cfg4.txt 2>&1; grep -v "^   (" $OUT/latency_cfg4.txt | tail -12
python tools/gpu_throughput.py cfg2 1 4 8 10 12 > $OUT/throughput.txt 2>&1; cat $OUT/throughput.txt

#!/bin/bash
# quick probe: parity subset, single-stream latency (cfg2, cfg4), saturated throughput, kernel durations under load
TAG=${1:-probe}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_codec_golden.py tests/test_delta_gpu.py tests/test_evaluate_app.py -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
python tools/gpu_latency.py cfg2 20 > $OUT/latency_cfg2.txt 2>&1; grep -v "^   (" $OUT/latency_cfg2.txt | tail -9
python tools/gpu_latency.py cfg3v 20 > $OUT/latency_cfg3v.txt 2>&1; grep -v "^   (" $OUT/latency_cfg3v.txt | tail -9
python tools/gpu_latency.py cfg1 20 > $OUT/latency_cfg1.txt 2>&1; grep -v "^   (" $OUT/latency_cfg1.txt | tail -9
python tools/gpu_throughput.py cfg3v 12 | tee $OUT/thr_cfg3v.txt

#!/bin/bash
# One GPU-box session of round 4, in stages so that a session can be cut into several gpurun calls and whatever time the box
# gives is spent on the most important evidence first.  The defaults at HEAD are the forms that HAVE run on an MI355X (round 2:
# two-kernel front end, one sort ticket counter, host entropy stage) plus round 3's layout-only changes; everything with a
# failure mode under load is opt-in and gets its A/B here.
#   bash tools/r04_round.sh <tag> <stage>      stage = parity | bench | ab | prof | all      -> gpurun_out/<tag>/
#     parity  the whole -m gpu suite + smoke at HEAD; if red: the same subset with each layout change off, the shfl build, and
#             the round-2 library (cwi-pcl-codec_amd/libpcc_hip_r02.so: the last one that was byte-green on the chip)
#     bench   bench.py at the driver's step count, the default, cfg4
#     ab      every optional form against the default: single-frame latency, saturated GPU stage, parity subset for both
#     prof    rocprofv3 kernel traces + PMC traffic (cfg2, cfg4, cfg2 with 12 distinct frames), the bench under the tracer
TAG=${1:-r04}; STAGE=${2:-all}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R02=$PWD/cwi-pcl-codec_amd/libpcc_hip_r02.so
SHFL=$PWD/cwi-pcl-codec_amd/libpcc_hip_shfl.so
want() { [ "$STAGE" = all ] || [ "$STAGE" = "$1" ]; }

if want parity; then
  python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -3 $OUT/pytest_gpu.log
  python -c "import __graft_entry__ as G; G.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
  if ! grep -q " passed" $OUT/pytest_gpu.log || grep -q " failed\| error" $OUT/pytest_gpu.log; then
    SUBSET="tests/test_gpu_parity.py tests/test_codec_golden.py"
    for SW in "PCC_LEAF_PROBES=uniform" "PCC_LEAF_ROWS=linear" "PCC_LEAF_PROBES=uniform PCC_LEAF_ROWS=linear"; do
      env $SW python -m pytest $SUBSET -m gpu -x -q > "$OUT/pytest_gpu_${SW// /_}.log" 2>&1
      echo "with $SW: $(tail -1 "$OUT/pytest_gpu_${SW// /_}.log")"
    done
    [ -f $SHFL ] && { PCC_LIB=$SHFL python -m pytest $SUBSET -m gpu -x -q > $OUT/pytest_gpu_shfl.log 2>&1; echo "shfl build: $(tail -1 $OUT/pytest_gpu_shfl.log)"; }
    # the round-2 library under HEAD's tests, without -x: what it never had (trees deeper than 21 levels, LINES on the GPU) fails, the rest says
    # whether the box and the tests are sound
    [ -f $R02 ] && { PCC_LIB=$R02 python -m pytest $SUBSET -m gpu -q -k "not 22_to_31" > $OUT/pytest_gpu_r02lib.log 2>&1; echo "round-2 library: $(tail -1 $OUT/pytest_gpu_r02lib.log)"; }
  fi
fi

if want bench; then
  python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "bench(20) rc=$?"; cat $OUT/bench_steps20.json
  python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
  python bench.py --workload cfg4 --steps 48 --warmup 4 --no-host-input > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; echo "bench(cfg4) rc=$?"; cat $OUT/bench_cfg4.json
fi

if want ab; then
  ab() { bash tools/ab_probe.sh $TAG/$1 $2 $3 $4 $5 > $OUT/$1.txt 2>&1; echo "---- $1"; tail -${6:-24} $OUT/$1.txt; }
  ab ab_fused PCC_FUSED_KEYS 1 - cfg2 40            # fused front end against the two-kernel default (watch the fallback chunks under ten frames in flight)
  ab ab_sortxcd PCC_SORT_XCD 16 - cfg2              # XCD-aware sort tickets against one counter
  ab ab_sortxcd_cfg4 PCC_SORT_XCD 16 - cfg4
  ab ab_probes PCC_LEAF_PROBES - uniform cfg2       # geometric first probes (default) against round 2's evenly spaced ones
  ab ab_sortlocal PCC_SORT_LOCAL 1 - cfg2           # local fix-up of the low code bits, one global sort pass fewer
  ab ab_sortlocal_cfg4 PCC_SORT_LOCAL 1 - cfg4
  ab ab_sortbare PCC_SORT_BARE 1 - cfg2             # payload-free sort passes, three tiles per CU
  [ -f $SHFL ] && ab ab_shfl PCC_LIB $PWD/cwi-pcl-codec_amd/libpcc_hip.so $SHFL cfg2
  [ -f $R02 ] && ab ab_r02lib PCC_LIB $PWD/cwi-pcl-codec_amd/libpcc_hip.so $R02 cfg2     # HEAD against the last library that ran on the chip
  # the host coder with sixteen frames per call (AVX-512 lanes) against four (scalar loop): value and host CPU per frame
  for BATCH in 4 16; do
    PCC_PIPELINE_BATCH=$BATCH python bench.py --steps 1024 --warmup 8 --no-cpu-baseline --no-host-input > $OUT/bench_batch_$BATCH.json 2> $OUT/bench_batch_$BATCH.err
    echo "frames per coder call $BATCH: $(python -c "import json,sys; d=json.load(open('$OUT/bench_batch_$BATCH.json')); print(d['value'], d['host_cpu_ms_per_frame'], d['entropy_stage'])" 2>&1)"
  done
  (cd tools/ubench && g++ -O3 -march=x86-64-v3 -I../../cwi-pcl-codec_amd/csrc -I../../include rc_many.cpp ../../cwi-pcl-codec_amd/csrc/pcc_host_codec.o -o rc_many && ./rc_many && PCC_RC_WIDE=0 ./rc_many | tail -4) > $OUT/rc_many.txt 2>&1; tail -24 $OUT/rc_many.txt
  # where the entropy stage of a long call should run on this box: host (default), GPU, the cost estimate
  for M in host gpu auto; do
    PCC_PIPELINE_ENTROPY=$M python bench.py --steps 1024 --warmup 8 --no-cpu-baseline --no-host-input > $OUT/bench_entropy_$M.json 2> $OUT/bench_entropy_$M.err
    echo "entropy=$M: $(python -c "import json,sys; d=json.load(open('$OUT/bench_entropy_$M.json')); print(d['value'], d['entropy_stage'])" 2>&1)"
  done
fi

if want prof; then
  for WL in cfg2 cfg4; do
    bash tools/prof_latency.sh $TAG/lat_$WL $WL > $OUT/latency_$WL.txt 2>&1; tail -16 $OUT/latency_$WL.txt
    bash tools/pmc_run.sh $TAG/pmc_$WL $WL > $OUT/pmc_$WL.log 2>&1; tail -3 $OUT/pmc_$WL.log
  done
  PMC_DISTINCT=12 bash tools/pmc_run.sh $TAG/pmc_cfg2_12frames cfg2 > $OUT/pmc_cfg2_12frames.log 2>&1; tail -3 $OUT/pmc_cfg2_12frames.log
  PCC_FUSED_KEYS=1 bash tools/pmc_run.sh $TAG/pmc_cfg2_fused cfg2 > $OUT/pmc_cfg2_fused.log 2>&1; tail -3 $OUT/pmc_cfg2_fused.log
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-input --steps 256 > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err)
  DB=$(find $OUT/prof -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/bench_kernel_stats.txt && cat $OUT/bench_kernel_stats.txt
  python tools/rc_device_speed.py > $OUT/rc_device_speed.txt 2>&1; tail -16 $OUT/rc_device_speed.txt   # both forms of the device range coder
  find $OUT -name '*.db' -size +20M -delete
fi

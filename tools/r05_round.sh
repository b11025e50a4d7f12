#!/bin/bash
# One GPU-box session, in stages so that it can be cut into several gpurun calls and whatever time the box gives is spent on
# the most important evidence first.  Nothing in cwi-pcl-codec_amd/csrc at HEAD has run on an MI355X since mid-round 2: the
# product is the forms that HAD run there (two-kernel front end, one sort ticket counter per pass, host entropy stage) plus
# layout-only changes; the opt-in forms of rounds 3-4 left the product in round 5 and come back here as a LIBRARY column.
#   bash tools/r05_round.sh <tag> <stage>      stage = parity | bench | prof | ab | all      -> gpurun_out/<tag>/
#     parity  the whole -m gpu suite + smoke at HEAD; if red, the bisect ladder on the parity subset: the developer build with
#             each layout change off, the shfl build, round 4's HEAD (libpcc_hip_r04x.so), the round-2 library (libpcc_hip_r02.so:
#             the last one that was byte-green on the chip)
#     bench   bench.py at the driver's step count, the default, cfg4
#     prof    rocprofv3 kernel traces + PMC traffic (cfg2, cfg4, cfg2 with 12 distinct frames), the bench under the tracer
#     ab      HEAD against the other libraries and switches: single-frame latency, saturated GPU stage, parity subset for both
TAG=${1:-r05}; STAGE=${2:-all}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
P=$PWD/cwi-pcl-codec_amd
HEAD_LIB=$P/libpcc_hip.so; DEV=$P/libpcc_hip_dev.so; SHFL=$P/libpcc_hip_shfl.so; R04X=$P/libpcc_hip_r04x.so; R02=$P/libpcc_hip_r02.so
want() { [ "$STAGE" = all ] || [ "$STAGE" = "$1" ]; }

if want parity; then
  python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -3 $OUT/pytest_gpu.log
  python -c "import __graft_entry__ as G; G.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
  if ! grep -q " passed" $OUT/pytest_gpu.log || grep -q " failed\| error" $OUT/pytest_gpu.log; then
    SUBSET="tests/test_gpu_parity.py tests/test_codec_golden.py"
    # without -x: which tests fail says where to look
    python -m pytest $SUBSET -m gpu -q --timeout 900 > $OUT/pytest_gpu_subset_head.log 2>&1; echo "HEAD, subset: $(tail -1 $OUT/pytest_gpu_subset_head.log)"
    if [ -f $DEV ]; then
      for SW in "PCC_LEAF_PROBES=uniform" "PCC_LEAF_ROWS=linear" "PCC_LEAF_PROBES=uniform PCC_LEAF_ROWS=linear" "PCC_SORT_SHAPE=wide" "PCC_WAIT=event"; do
        env PCC_LIB=$DEV $SW python -m pytest $SUBSET -m gpu -q --timeout 900 > "$OUT/pytest_gpu_dev_${SW// /_}.log" 2>&1
        echo "developer build with $SW: $(tail -1 "$OUT/pytest_gpu_dev_${SW// /_}.log")"
      done
    fi
    [ -f $SHFL ] && { PCC_LIB=$SHFL python -m pytest $SUBSET -m gpu -q --timeout 900 > $OUT/pytest_gpu_shfl.log 2>&1; echo "shfl build: $(tail -1 $OUT/pytest_gpu_shfl.log)"; }
    # older libraries under HEAD's tests: what they never had fails (r02: trees deeper than 21 levels, LINES on the GPU), the rest
    # says whether the box and the tests are sound
    [ -f $R04X ] && { PCC_LIB=$R04X python -m pytest $SUBSET -m gpu -q --timeout 900 > $OUT/pytest_gpu_r04x.log 2>&1; echo "round 4's HEAD: $(tail -1 $OUT/pytest_gpu_r04x.log)"; }
    [ -f $R02 ] && { PCC_LIB=$R02 python -m pytest $SUBSET -m gpu -q --timeout 900 -k "not 22_to_31" > $OUT/pytest_gpu_r02.log 2>&1; echo "round-2 library: $(tail -1 $OUT/pytest_gpu_r02.log)"; }
  fi
fi

if want bench; then
  python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "bench(20) rc=$?"; cat $OUT/bench_steps20.json
  python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
  python bench.py --workload cfg4 --steps 48 --warmup 4 --no-host-input > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; echo "bench(cfg4) rc=$?"; cat $OUT/bench_cfg4.json
fi

if want prof; then
  for WL in cfg2 cfg4; do
    bash tools/prof_latency.sh $TAG/lat_$WL $WL > $OUT/latency_$WL.txt 2>&1; tail -16 $OUT/latency_$WL.txt
    bash tools/pmc_run.sh $TAG/pmc_$WL $WL > $OUT/pmc_$WL.log 2>&1; tail -3 $OUT/pmc_$WL.log
  done
  PMC_DISTINCT=12 bash tools/pmc_run.sh $TAG/pmc_cfg2_12frames cfg2 > $OUT/pmc_cfg2_12frames.log 2>&1; tail -3 $OUT/pmc_cfg2_12frames.log
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-input --steps 256 > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err)
  DB=$(find $OUT/prof -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/bench_kernel_stats.txt && cat $OUT/bench_kernel_stats.txt
  find $OUT -name '*.db' -size +20M -delete
fi

if want ab; then
  export PCC_ALLOW_NON_PRODUCT_LIB=1
  ab() { bash tools/ab_probe.sh $TAG/$1 "$2" "$3" $4 > $OUT/$1.txt 2>&1; echo "---- $1"; tail -${5:-24} $OUT/$1.txt; }
  # HEAD against the libraries of older commits (the second setting of each pair is "KEY=value ..." for env)
  [ -f $R02 ] && ab ab_r02 "PCC_LIB=$HEAD_LIB" "PCC_LIB=$R02" cfg2                        # the last library that ran on the chip
  [ -f $R04X ] && ab ab_r04x "PCC_LIB=$HEAD_LIB" "PCC_LIB=$R04X" cfg2                     # round 4's HEAD, its forms off: what the removals cost or gained
  [ -f $SHFL ] && ab ab_shfl "PCC_LIB=$HEAD_LIB" "PCC_LIB=$SHFL" cfg2                     # DPP wave scans against ds_bpermute ones
  if [ -f $R04X ]; then                                                                   # the forms that left the product, each against round 4's defaults
    ab ab_fused "PCC_LIB=$R04X" "PCC_LIB=$R04X PCC_FUSED_KEYS=1" cfg2 40
    ab ab_sortlocal "PCC_LIB=$R04X" "PCC_LIB=$R04X PCC_SORT_LOCAL=1" cfg2
    ab ab_sortlocal_cfg4 "PCC_LIB=$R04X" "PCC_LIB=$R04X PCC_SORT_LOCAL=1" cfg4
    ab ab_sortbare "PCC_LIB=$R04X" "PCC_LIB=$R04X PCC_SORT_BARE=1" cfg2
  fi
  if [ -f $DEV ]; then                                                                    # round 3's layouts against round 2's
    ab ab_probes "PCC_LIB=$DEV" "PCC_LIB=$DEV PCC_LEAF_PROBES=uniform" cfg2
    ab ab_rows "PCC_LIB=$DEV" "PCC_LIB=$DEV PCC_LEAF_ROWS=linear" cfg2
  fi
  unset PCC_ALLOW_NON_PRODUCT_LIB
  # the host coder with sixteen frames per call (AVX-512 lanes) against four (scalar loop): value and host CPU per frame
  for BATCH in 4 16; do
    PCC_PIPELINE_BATCH=$BATCH python bench.py --steps 1024 --warmup 8 --no-cpu-baseline --no-host-input > $OUT/bench_batch_$BATCH.json 2> $OUT/bench_batch_$BATCH.err
    echo "frames per coder call $BATCH: $(python -c "import json,sys; d=json.load(open('$OUT/bench_batch_$BATCH.json')); print(d['value'], d['host_cpu_ms_per_frame'], d['entropy_stage'])" 2>&1)"
  done
  (cd tools/ubench && g++ -O3 -march=x86-64-v3 -I../../cwi-pcl-codec_amd/csrc -I../../include rc_many.cpp ../../cwi-pcl-codec_amd/csrc/dev_obj/pcc_host_codec.o -o rc_many && ./rc_many && PCC_RC_WIDE=0 ./rc_many | tail -4) > $OUT/rc_many.txt 2>&1; tail -24 $OUT/rc_many.txt
  # where the entropy stage of a long call should run on this box: host (default) or GPU
  for M in host gpu; do
    PCC_PIPELINE_ENTROPY=$M python bench.py --steps 1024 --warmup 8 --no-cpu-baseline --no-host-input > $OUT/bench_entropy_$M.json 2> $OUT/bench_entropy_$M.err
    echo "entropy=$M: $(python -c "import json,sys; d=json.load(open('$OUT/bench_entropy_$M.json')); print(d['value'], d['entropy_stage'])" 2>&1)"
  done
  python tools/rc_device_speed.py > $OUT/rc_device_speed.txt 2>&1; tail -16 $OUT/rc_device_speed.txt   # both forms of the device range coder
fi

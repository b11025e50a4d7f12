#!/bin/bash
# One GPU-box session of round 3, ordered so that whatever time the box gives is spent on the most important evidence first:
#   1 parity: the whole -m gpu suite + smoke at HEAD (nothing else counts without it)
#   2 bench lines: the driver's step count, the default, cfg4
#   3 A/B of what round 3 built blind: fused keys on/off (latency + saturated GPU stage, parity for both), the shfl build
#   4 single-stream rocprofv3 kernel traces + PMC traffic, cfg2 and cfg4, and cfg2 with 12 distinct frames (HBM, not the
#     Infinity Cache)
#   5 rocprofv3 kernel stats of the bench command itself; device range coder speed
# Usage (through gpurun): bash tools/r03_round.sh <tag> [quick]      -> gpurun_out/<tag>/ ; copy the summaries to profiles/
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
if ! grep -q " passed" $OUT/pytest_gpu.log || grep -q " failed" $OUT/pytest_gpu.log; then
  # something fails on the chip that passed on the executor: which of round 3's restructurings is it?  The same tests with each
  # one switched off in turn, then with all of them off (round 2's form of the kernels)
  for SW in "PCC_FUSED_KEYS=0" "PCC_SORT_XCD=0" "PCC_LEAF_PROBES=uniform" "PCC_FUSED_KEYS=0 PCC_SORT_XCD=0 PCC_LEAF_PROBES=uniform"; do
    env $SW python -m pytest tests/test_gpu_parity.py tests/test_codec_golden.py -m gpu -x -q -k "not two_kernel_form" > "$OUT/pytest_gpu_${SW// /_}.log" 2>&1
    echo "with $SW: $(tail -1 "$OUT/pytest_gpu_${SW// /_}.log")"
  done
  [ -f cwi-pcl-codec_amd/libpcc_hip_shfl.so ] && { PCC_LIB=$PWD/cwi-pcl-codec_amd/libpcc_hip_shfl.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not two_kernel_form" > $OUT/pytest_gpu_shfl.log 2>&1; echo "shfl build: $(tail -1 $OUT/pytest_gpu_shfl.log)"; }
fi
python -c "import __graft_entry__ as G; G.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "bench(20) rc=$?"; cat $OUT/bench_steps20.json
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
python bench.py --workload cfg4 --steps 48 --warmup 4 --no-host-input > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; echo "bench(cfg4) rc=$?"; cat $OUT/bench_cfg4.json
[ "$2" == "quick" ] && exit 0
# ---- A/B: fused keys (the parity subset runs for both settings inside ab_probe)
bash tools/ab_probe.sh $TAG/ab_fused PCC_FUSED_KEYS 1 0 cfg2 > $OUT/ab_fused.txt 2>&1; tail -40 $OUT/ab_fused.txt
# ---- A/B: first probes of k_leaf_tile's parent search (geometric, the default, against round 2's evenly spaced ones)
bash tools/ab_probe.sh $TAG/ab_probes PCC_LEAF_PROBES - uniform cfg2 > $OUT/ab_probes.txt 2>&1; tail -24 $OUT/ab_probes.txt
# ---- A/B: XCD-aware sort tickets (default: chunks of 16 tiles per XCD) against one counter for all workgroups
bash tools/ab_probe.sh $TAG/ab_sortxcd PCC_SORT_XCD 16 0 cfg2 > $OUT/ab_sortxcd.txt 2>&1; tail -24 $OUT/ab_sortxcd.txt
bash tools/ab_probe.sh $TAG/ab_sortxcd4 PCC_SORT_XCD 16 0 cfg4 > $OUT/ab_sortxcd_cfg4.txt 2>&1; tail -24 $OUT/ab_sortxcd_cfg4.txt
# ---- A/B: payload-free sort passes, three tiles per CU (an experiment, off by default)
bash tools/ab_probe.sh $TAG/ab_sortbare PCC_SORT_BARE - 1 cfg2 > $OUT/ab_sortbare.txt 2>&1; tail -24 $OUT/ab_sortbare.txt
# ---- A/B: local fix-up of the low code bits in the leaf scan, one global sort pass fewer (an experiment, off by default)
bash tools/ab_probe.sh $TAG/ab_sortlocal PCC_SORT_LOCAL - 1 cfg2 > $OUT/ab_sortlocal.txt 2>&1; tail -24 $OUT/ab_sortlocal.txt
bash tools/ab_probe.sh $TAG/ab_sortlocal4 PCC_SORT_LOCAL - 1 cfg4 > $OUT/ab_sortlocal_cfg4.txt 2>&1; tail -24 $OUT/ab_sortlocal_cfg4.txt
# ---- the shfl build of the wave helpers against the DPP one
if [ -f cwi-pcl-codec_amd/libpcc_hip_shfl.so ]; then
  bash tools/ab_probe.sh $TAG/ab_shfl PCC_LIB $PWD/cwi-pcl-codec_amd/libpcc_hip.so $PWD/cwi-pcl-codec_amd/libpcc_hip_shfl.so cfg2 > $OUT/ab_shfl.txt 2>&1; tail -24 $OUT/ab_shfl.txt
fi
# ---- single-stream kernel traces and HBM traffic
for WL in cfg2 cfg4; do
  bash tools/prof_latency.sh $TAG/lat_$WL $WL > $OUT/latency_$WL.txt 2>&1; tail -16 $OUT/latency_$WL.txt
  bash tools/pmc_run.sh $TAG/pmc_$WL $WL > $OUT/pmc_$WL.log 2>&1; tail -3 $OUT/pmc_$WL.log
done
PCC_FUSED_KEYS=0 PCC_LEAF_PROBES=uniform PCC_SORT_XCD=0 bash tools/pmc_run.sh $TAG/pmc_cfg2_round2_form cfg2 > $OUT/pmc_cfg2_round2_form.log 2>&1; tail -3 $OUT/pmc_cfg2_round2_form.log
PMC_DISTINCT=12 bash tools/pmc_run.sh $TAG/pmc_cfg2_12frames cfg2 > $OUT/pmc_cfg2_12frames.log 2>&1; tail -3 $OUT/pmc_cfg2_12frames.log
# ---- the bench command under the kernel tracer; the device range coder
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-input --steps 256 > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err)
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/bench_kernel_stats.txt && cat $OUT/bench_kernel_stats.txt
python tools/rc_device_speed.py > $OUT/rc_device_speed.txt 2>&1; tail -5 $OUT/rc_device_speed.txt
find $OUT -name '*.db' -size +20M -delete

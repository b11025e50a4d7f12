#!/bin/bash
# After a GPU session: copy the summaries of gpurun_out/<tag>/ (scratch, untracked) that are quoted as evidence into
# profiles/ under the round's name, so that nothing judged depends on a scratch directory.  (CPU; no GPU needed.)
#   bash tools/collect_profiles.sh <tag> [<round prefix, default r05>]
TAG=${1:?usage: collect_profiles.sh <tag> [prefix]}; P=${2:-r05}
S=gpurun_out/$TAG; D=profiles
[ -d $S ] || { echo "no $S"; exit 1; }
cpif() { [ -s "$1" ] && cp "$1" "$2" && echo "  $2"; }
echo "from $S:"
cpif $S/pytest_gpu.log                          $D/${P}_pytest_gpu.log
cpif $S/smoke.log                               $D/${P}_smoke.log
cpif $S/bench_steps20.json                      $D/${P}_bench_steps20.json
cpif $S/bench.json                              $D/${P}_bench.json
cpif $S/bench_cfg4.json                         $D/${P}_bench_cfg4.json
for WL in cfg2 cfg4; do
  cpif $S/lat_$WL/kernel_stats.txt              $D/${P}_${WL}_single_stream_kernel_stats.txt
  cpif $S/lat_$WL/latency.txt                   $D/${P}_${WL}_single_stream_latency.txt
  cpif $S/pmc_$WL/hbm_traffic_$WL.json          $D/${P}_hbm_traffic_$WL.json
done
cpif $S/pmc_cfg2_12frames/hbm_traffic_cfg2.json $D/${P}_hbm_traffic_cfg2_12_distinct_frames.json
cpif $S/bench_kernel_stats.txt                  $D/${P}_bench_kernel_stats.txt
cpif $S/rc_device_speed.txt                     $D/${P}_rc_device_speed.txt
cpif $S/rc_many.txt                             $D/${P}_rc_many_host_coder.txt
for f in $S/ab_*.txt $S/bench_entropy_*.json $S/bench_batch_*.json; do [ -s "$f" ] && cpif $f $D/${P}_$(basename $f); done
# the bench line's traffic figure comes from profiles/hbm_traffic_<workload>.json: a fresh one replaces round 2's
for WL in cfg2 cfg4; do [ -s $S/pmc_$WL/hbm_traffic_$WL.json ] && cp $S/pmc_$WL/hbm_traffic_$WL.json $D/hbm_traffic_$WL.json && echo "  $D/hbm_traffic_$WL.json (replaced)"; done

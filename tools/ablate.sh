#!/bin/bash
# Profiling helper (GPU box): rocprofv3 kernel trace of the bench under different PCC_ABLATE masks.
cd /tmp && export TMPDIR=/tmp
for m in "$@"; do
  rm -rf /tmp/abl_$m
  PCC_ABLATE=$m rocprofv3 --kernel-trace -d /tmp/abl_$m -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --workers 1 > /dev/null 2>&1
  echo "== PCC_ABLATE=$m"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/abl_$m/t_results.db | grep -E "k_leaf_finalize"
done

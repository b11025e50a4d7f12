#!/bin/bash
# Waits for the GPU gate.  Every PERIOD seconds: one gpurun call of the parity stage only.  A refused call costs nothing and returns at
# once; the first call that is NOT refused ends the loop whatever it says, so that somebody reads the first byte comparison before any
# further GPU-minute is spent (advisor, round 5: no unattended escalation to prof / ab).
#   bash tools/gpu_poll.sh <tag> [period_s] [max_tries]      history -> notes/<tag>_poll_history.txt
TAG=${1:-r06}; PERIOD=${2:-600}; MAX=${3:-70}
HIST=notes/${TAG}_poll_history.txt
for i in $(seq 1 $MAX); do
  /usr/local/graft/bin/gpurun --timeout 2700 -- "bash tools/r05_round.sh $TAG parity" > /tmp/gpu_poll_last.log 2>&1
  ST=$(python3 -c "import json; d=json.load(open('gpurun_out/.last_call.json')); print(d.get('status'), d.get('fault'), d.get('rc'))" 2>/dev/null)
  echo "$(date -u +%FT%TZ) try $i: $ST" >> $HIST
  case "$ST" in refused*|busy*|"no_box"*|"") sleep $PERIOD;; *) echo "call went through: $ST"; tail -40 /tmp/gpu_poll_last.log; exit 0;; esac
done
echo "gate stayed closed for $MAX tries"; exit 3

#!/bin/bash
# Round 6, VERDICT r5 item 1b: the overlapped two-graph step held to ONE bit pattern.
#   (1) the SHIPPED mode (--overlap, forks in graph A only) and the fork-free two-graph step: 1000 replays, ten processes
#   (2) the known-bad layout (--overlap-forks b --unsafe) alone, then with the round's candidate mechanisms removed
# Output: gpurun_out/r06_twograph.txt (copied to profiles/r06/two_graph_repeat_check.txt)
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_twograph.txt
mkdir -p gpurun_out
: > $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # label, args...
  local label=$1; shift
  echo "== $label: $*" >> $OUT
  timeout 600 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --warmup 1 --head --overlap "$@" 2>/tmp/err.txt \
    | grep '^{' | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print(json.dumps({k:l[k] for k in ('repeat_check','forks','debug','graph','distinct_late','distinct_early','varying_parameters') if k in l}))" >> $OUT 2>&1 \
    || { echo "FAILED" >> $OUT; tail -5 /tmp/err.txt >> $OUT; }
}
R=${R:-1000}
for i in 1 2 3 4 5 6 7 8 9 10; do run "shipped a #$i" --overlap-forks a --repeat-check $R; done
for i in 1 2 3; do run "none #$i" --overlap-forks none --repeat-check $R; done
run "bad b" --overlap-forks b --unsafe --repeat-check 200
run "bad b same_stream" --overlap-forks b --unsafe --repeat-check 200 --debug-two-graphs same_stream
run "bad b same_stream #2" --overlap-forks b --unsafe --repeat-check 200 --debug-two-graphs same_stream
run "bad both same_stream" --overlap-forks both --unsafe --repeat-check 200 --debug-two-graphs same_stream
run "a same_stream" --overlap-forks a --repeat-check $R --debug-two-graphs same_stream
echo "== done" >> $OUT
cat $OUT

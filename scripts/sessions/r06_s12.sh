#!/bin/bash
# Round 6 session 12: where a repeat-check run of the two-graph step spends its time (stack dumps of both ranks)
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s12
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 20 60; do
  echo "== repeat-check $n" | tee -a $OUT/summary.txt
  ( time timeout 150 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --warmup 1 --head --overlap --overlap-forks none --repeat-check $n > $OUT/rc_$n.out 2> $OUT/rc_$n.err ) 2>&1 | grep real | tee -a $OUT/summary.txt
  grep '^{' $OUT/rc_$n.out | tail -1 | cut -c1-300 | tee -a $OUT/summary.txt
done
echo "== repeat-check 200 with stack dumps after 60 and 90 s" | tee -a $OUT/summary.txt
timeout 150 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --warmup 1 --head --overlap --overlap-forks none --repeat-check 200 > $OUT/rc_200.out 2> $OUT/rc_200.err &
PID=$!
sleep 60
for c in $(pgrep -P $PID) ; do for cc in $(pgrep -P $c) $c; do kill -USR1 $cc 2>/dev/null; done; done
sleep 30
for c in $(pgrep -P $PID) ; do for cc in $(pgrep -P $c) $c; do kill -USR1 $cc 2>/dev/null; done; done
wait $PID
grep '^{' $OUT/rc_200.out | tail -1 | cut -c1-300 | tee -a $OUT/summary.txt
grep -n "File \"/root/repo\|File \".*bench_backbone\|Thread\|most recent" $OUT/rc_200.err | head -60 | cut -c1-200 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

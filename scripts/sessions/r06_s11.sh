#!/bin/bash
# Round 6 session 11: why session 10's repeat-check runs sat until their timeouts; the pipelined-pair experiment
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s11
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== one repeat-check run, 200 replays, stderr kept" | tee $OUT/summary.txt
( time timeout 240 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --warmup 1 --head --overlap --overlap-forks none --repeat-check 200 > $OUT/rc_stdout.txt 2> $OUT/rc_stderr.txt ) 2>&1 | tail -4 | tee -a $OUT/summary.txt
echo "rc=$?" | tee -a $OUT/summary.txt
grep '^{' $OUT/rc_stdout.txt | tail -1 | cut -c1-400 | tee -a $OUT/summary.txt
tail -8 $OUT/rc_stderr.txt | cut -c1-300 | tee -a $OUT/summary.txt
echo "== bench.py --pipelined" | tee -a $OUT/summary.txt
for i in 1 2; do
  timeout 200 python bench.py --pipelined --no-cpu-baseline --no-kernel-roofline --backbone off 2>$OUT/pipe_err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('headline', d['ms_per_step'], d['value'], '| pipelined pair', d.get('pipelined_pair'))" | cut -c1-400 | tee -a $OUT/summary.txt
  tail -2 $OUT/pipe_err.txt | cut -c1-300 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
OUT=gpurun_out/r05g
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for cfg in "" "--overlap --overlap-forks none" "--overlap --overlap-forks a"; do
  timeout 300 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --warmup 1 --head $cfg --repeat-check 200 2>$OUT/err.txt | grep repeat_check | python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['varying_parameters']
print('[$cfg]', 'late', d['distinct_late'][:4], 'early', d['distinct_early'][:4], 'varying parameters', len(v), list(v)[:6])" | tee -a $OUT/summary.txt
done
done
tail -5 $OUT/err.txt

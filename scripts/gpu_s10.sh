#!/bin/bash
OUT=gpurun_out/r04p
mkdir -p $OUT
export TMPDIR=/tmp
echo "== graph B's fork episodes and their shapes" | tee $OUT/summary.txt
timeout 600 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --warmup 1 --steps 1 --head --overlap --overlap-forks b --debug-two-graphs trace_scratch 2>&1 | grep -E "scratch|ptr|Error" | cut -c1-200 | head -14 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# Round 6 session 25: does the bf16 network learn like its f32 twin?  1000 SGD steps x 3 streams of config 2 + classifier
# on a synthetic 8-class task, both precisions, same batches / parameters / dropout masks
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s25
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python scripts/bf16_loss_curve.py --steps 1000 --seeds 3 > $OUT/bf16_loss_curve.json 2> $OUT/err.log
echo "rc=$?" | tee $OUT/summary.txt
tail -5 $OUT/err.log | cut -c1-300 | tee -a $OUT/summary.txt
cat $OUT/bf16_loss_curve.json | cut -c1-3000 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

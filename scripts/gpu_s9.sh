#!/bin/bash
OUT=gpurun_out/r04i
mkdir -p $OUT
export TMPDIR=/tmp
echo "== support-pass variants" | tee $OUT/summary.txt
timeout 1500 python scripts/micro/kernel_variants.py --run --step 2>&1 | cut -c1-400 | tee $OUT/variants.jsonl | tee -a $OUT/summary.txt
echo "== two graphs, forks in graph B: what makes the early-stage gradients right" | tee -a $OUT/summary.txt
for dbg in "" "sync_between" "own_pool" "fresh_streams" "eager_b" "sync_between,own_pool"; do
  echo "-- --overlap --overlap-forks b --debug-two-graphs '$dbg'" | tee -a $OUT/summary.txt
  CL3D_DP_DEBUG=1 timeout 600 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --steps 2 --warmup 3 --checksums --head --overlap --overlap-forks b --debug-two-graphs "$dbg" 2>&1 | grep -E "debug step|Error|error|Segmentation|Fatal" | cut -c1-200 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

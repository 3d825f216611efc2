#!/bin/bash
OUT=gpurun_out/r04g
mkdir -p $OUT
export TMPDIR=/tmp
echo "== support pass: definition tests" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_pwmlp_support_gpu.py tests/test_operators_gpu.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -5 | tee -a $OUT/summary.txt
echo "== variants" | tee -a $OUT/summary.txt
timeout 1500 python scripts/micro/kernel_variants.py --run --step 2>&1 | tee $OUT/variants.jsonl | tee -a $OUT/summary.txt
echo "== two graphs with the forks on (2 ranks on one device over gloo): per-step gradient norms, flat / overlap / overlap with forks" | tee -a $OUT/summary.txt
for flags in "" "--overlap" "--overlap --overlap-forks"; do
  echo "-- flags: $flags" | tee -a $OUT/summary.txt
  CL3D_DP_DEBUG=1 timeout 600 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --steps 3 --warmup 4 --checksums --head $flags 2>&1 | grep -E "^\{|debug step|Error|error|Segmentation|Fatal" | cut -c1-400 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

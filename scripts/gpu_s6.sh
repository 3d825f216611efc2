#!/bin/bash
OUT=gpurun_out/r05h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bottleneck_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25 | cut -c1-300 | tee $OUT/summary.txt
for c in s3dis_pseudogrid partnet_adaptive s3dis_pospool_deep; do
  for rep in 1 2; do
  timeout 600 python scripts/bench_backbone.py --config $c 2>/dev/null | tail -1 | cut -c1-200 | sed 's/^/fused bottlenecks: /' | tee -a $OUT/summary.txt
  timeout 600 python scripts/bench_backbone.py --config $c --layerwise 2>/dev/null | tail -1 | cut -c1-200 | sed 's/^/layer by layer:    /' | tee -a $OUT/summary.txt
  done
done

#!/bin/bash
OUT=gpurun_out/r04h
mkdir -p $OUT
export TMPDIR=/tmp
echo "== two graphs (2 ranks on one device over gloo): which capture's forks break the early-stage gradients" | tee $OUT/summary.txt
for flags in "--overlap" "--overlap --overlap-forks a" "--overlap --overlap-forks b" "--overlap --overlap-forks both" "--overlap --overlap-forks both --lead-kernel" "--overlap --overlap-forks b --lead-kernel"; do
  echo "-- flags: $flags" | tee -a $OUT/summary.txt
  CL3D_DP_DEBUG=1 timeout 600 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --steps 2 --warmup 4 --checksums --head $flags 2>&1 | grep -E "debug step|Error|error|Segmentation|Fatal" | cut -c1-200 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

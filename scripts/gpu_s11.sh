#!/bin/bash
OUT=gpurun_out/r04r
mkdir -p $OUT
export TMPDIR=/tmp
echo "== eager launches: index streams off (default in eager mode) / forced on" | tee $OUT/summary.txt
for v in "CL3D_ASYNC=0" "CL3D_ASYNC=1"; do
  env $v timeout 300 python bench.py --no-graph --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms_per_step', d['ms_per_step'])" | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

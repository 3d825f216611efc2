#!/bin/bash
OUT=gpurun_out/r05j
mkdir -p $OUT
export TMPDIR=/tmp
for c in modelnet_pointwisemlp s3dis_pseudogrid partnet_adaptive; do
  for v in "CL3D_CSR_FIRST=1" "CL3D_CSR_FIRST=0" "CL3D_CSR_FIRST=1" "CL3D_CSR_FIRST=0"; do
    env $v timeout 600 python scripts/bench_backbone.py --config $c 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', '$v', d['ms_per_step'], d['launch'])" | tee -a $OUT/summary.txt
  done
done
timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | tail -1 | cut -c1-230 | tee -a $OUT/summary.txt
bm() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms_per_step', d['ms_per_step'], d['config']['launch'])"; }
timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "headline" | tee -a $OUT/summary.txt

#!/bin/bash
OUT=gpurun_out/r05k
mkdir -p $OUT
export TMPDIR=/tmp
for rate in 0 6e6 3e6; do
  echo "#### CL3D_GEMM_L2_RATE=$rate" | tee -a $OUT/summary.txt
  CL3D_GEMM_L2_RATE=$rate timeout 600 python scripts/bench_point_gemm.py --convs --reps 20 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); c=d['conv']; print(c['C'],c['Co'],c['N'],'f32',d['f32'],'bf16',d['bf16'])" | tee -a $OUT/summary.txt
  for prec in bf16 f32; do
    CL3D_GEMM_L2_RATE=$rate timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision $prec 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config2', '$prec', d['ms_per_step'])" | tee -a $OUT/summary.txt
  done
done

#!/bin/bash
# Round 6 session 15: tile / K-split sweep of the config-2 products under the XCD-aware tile order (is the planner's fit,
# made under the plain order, still choosing well?)
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s15
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python scripts/micro/gemm_plan_sweep.py --run --precisions bf16 > $OUT/gemm_plan_sweep_bf16_xcd.jsonl 2> $OUT/sweep.err
timeout 600 python scripts/micro/gemm_plan_sweep.py --run --precisions bf16 --point > $OUT/gemm_plan_sweep_bf16_point_xcd.jsonl 2>> $OUT/sweep.err
timeout 600 python scripts/micro/gemm_plan_sweep.py --run --precisions f32 > $OUT/gemm_plan_sweep_f32_xcd.jsonl 2>> $OUT/sweep.err
wc -l $OUT/*.jsonl | tee $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

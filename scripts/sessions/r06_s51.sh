#!/bin/bash
# Round 6 session 51: session 50 -- the gather pass's outputs vary beside a second process running the bf16 backbone, not beside the f32
# one.  Which of the second process's kernels: the contractions alone, by precision, kind and shape.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s51}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 SKIP_ALONE=1 VICTIMS=1 VICTIM_PRECS=1 REPS=120
export LOADS=${LOADS:-gemm_bf16_all_0_17,gemm_f32_all_0_17,gemm_bf16_conv_0_10,gemm_bf16_rows_0_7,gemm_bf16_conv_0_5,gemm_bf16_conv_5_10}
timeout 500 python scripts/micro/pwmlp_repeat_under_load.py 2>$OUT/err.log | cut -c1-250 | tee $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

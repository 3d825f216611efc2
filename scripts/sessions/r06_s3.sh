#!/bin/bash
# Round 6 session 3: in-launch slice sums without fences (device-scope stores / loads)
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s3
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_mfma_gemm_gpu.py tests/test_bq_tune.py tests/test_bq_paths_gpu.py tests/test_abi_host_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -6 | tee -a $OUT/summary.txt
echo "== config-2 backbone" | tee -a $OUT/summary.txt
for pr in bf16 f32; do timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision $pr 2>/dev/null | grep '^{' | tail -1 | cut -c1-300 | tee -a $OUT/summary.txt; done
echo "== plan sweep bf16 in-launch (no fences)" | tee -a $OUT/summary.txt
timeout 900 python scripts/micro/gemm_plan_sweep.py --run --precisions bf16 --fused-sum 1 > $OUT/gemm_plan_sweep_bf16_inlaunch.jsonl 2>$OUT/sweep.err
timeout 900 python scripts/micro/gemm_plan_sweep.py --run --precisions f32 --fused-sum 1 > $OUT/gemm_plan_sweep_f32_inlaunch.jsonl 2>>$OUT/sweep.err
echo "== ball query, tuned dispatch" | tee -a $OUT/summary.txt
for m in 1.5 4.0; do timeout 120 python scripts/bench_bq.py --mult $m --tuned | tee -a $OUT/summary.txt; for p in tile cells; do CL3D_BQ_PATH=$p timeout 120 python scripts/bench_bq.py --mult $m | tee -a $OUT/summary.txt; done; done
echo "== done" | tee -a $OUT/summary.txt

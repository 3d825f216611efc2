#!/bin/bash
# Round 6 session 49: the bf16 step under two ranks on one device varies from replay to replay (sessions 37-48; the gather pass reads
# rows that differ from what the copy engine reads at the same addresses).  Is it the in-launch K-slice sum of the producing contraction?
# Variant library (-DCL3D_GEMM_PLAN_ENV): CL3D_GEMM_FUSED_SUM=1 (shipped behaviour) against 0 (every slice sum a launch of its own).
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s49}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 CL3D_BENCH_ONE_DEVICE=1 CL3D_DP_NOEXCHANGE=1
export CL3D_LIB=$PWD/scripts/micro/var/libcl3d_gemm_plan_env.so
echo "== replays of one eager step, bf16, two ranks on one device" | tee $OUT/summary.txt
for f in 1 0 1 0; do
  CL3D_GEMM_FUSED_SUM=$f timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --gpus 2 --warmup 1 --gemm-plans model --weight-grads joined --repeat-check 40 --no-graph --precision bf16 2>>$OUT/err.log | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('fused_sum=$f', 'distinct late', d['distinct_late'][:6], 'early', d['distinct_early'][:6], '|', len(d['varying_parameters']), 'parameters vary')" | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

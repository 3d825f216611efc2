#!/bin/bash
# Round 6 session 44: session 42 with every autograd Function of the engine traced: the first OUTPUT that varies over the replays
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s44}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 CL3D_BENCH_ONE_DEVICE=1 CL3D_DP_NOEXCHANGE=1 CL3D_TRACE_PWMLP=${CL3D_TRACE_PWMLP:-0}
for i in 1 2; do
timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --gpus 2 --warmup 1 --gemm-plans model --weight-grads joined --repeat-check 24 --no-graph --dump-forward x --precision bf16 2>$OUT/bf16_$i.err | grep '^{' | tail -1 > $OUT/bf16_$i.json
grep varying_forward $OUT/bf16_$i.err | cut -c1-1500 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt
grep -h "first layer" $OUT/bf16_*.err | cut -c1-900 | head -24 | tee -a $OUT/summary.txt

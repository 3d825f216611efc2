#!/bin/bash
# Round 6 session 39: is the run-to-run variation of the two-rank one-device stand-in (session 38) rounding-level noise amplified by the
# bf16 rounding of the contractions' inputs?  The same runs in f32: norms should then agree to ~6 digits over the first steps.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s39}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 CL3D_BENCH_ONE_DEVICE=1 CL3D_DP_DEBUG=1
run() { # name, extra args
  local name=$1; shift
  timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --gpus 2 --steps 2 --warmup 8 --gemm-plans model "$@" 2>&1 | grep "debug step" > $OUT/$name.txt
  echo "-- $name" | tee -a $OUT/summary.txt; head -8 $OUT/$name.txt | tee -a $OUT/summary.txt
}
echo "== per-step norms, two ranks on one device, f32" | tee $OUT/summary.txt
run f32_graph_joined_1 --weight-grads joined; run f32_graph_joined_2 --weight-grads joined
run f32_graph_deferred_1 --weight-grads deferred; run f32_graph_deferred_2 --weight-grads deferred
run f32_eager_1 --no-graph; run f32_eager_2 --no-graph
echo "== done" | tee -a $OUT/summary.txt

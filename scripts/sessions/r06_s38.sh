#!/bin/bash
# Round 6 session 38: session 37 found the two-rank one-device stand-in varying from run to run in BOTH weight-gradient modes
# (one process: bit-equal).  Where does it start: per-step norms (CL3D_DP_DEBUG) of two runs each of graph / eager / graph with
# device-wide waits around the exchange.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s38}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 CL3D_BENCH_ONE_DEVICE=1 CL3D_DP_DEBUG=1
run() { # name, extra args
  local name=$1; shift
  timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --gpus 2 --steps 2 --warmup 8 --gemm-plans model --weight-grads joined "$@" 2>&1 | grep "debug step" > $OUT/$name.txt
  echo "-- $name" | tee -a $OUT/summary.txt; head -8 $OUT/$name.txt | tee -a $OUT/summary.txt
}
echo "== per-step norms, two ranks on one device" | tee $OUT/summary.txt
run graph_1; run graph_2
run eager_1 --no-graph; run eager_2 --no-graph
CL3D_DP_SYNC=1 run graph_sync_1; CL3D_DP_SYNC=1 run graph_sync_2
echo "== done" | tee -a $OUT/summary.txt

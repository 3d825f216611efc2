#!/bin/bash
# Round 6 session 40: the bf16 step under two ranks on one device is not bit-repeatable (sessions 37-39; f32 is).  Which parameters'
# gradients vary over replays of the SAME step (no optimizer): the deepest layer whose gradient varies names the kernel.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s40}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 CL3D_BENCH_ONE_DEVICE=1
run() { # name, args
  local name=$1; shift
  timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --gpus 2 --warmup 1 --gemm-plans model --weight-grads joined --repeat-check 40 "$@" 2>>$OUT/err.log | grep '^{' | tail -1 > $OUT/$name.json
  python - $OUT/$name.json "$name" <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
except Exception as e:
    print("--", sys.argv[2], "no line:", e); sys.exit(0)
v = d.get("varying_parameters", {})
print("--", sys.argv[2], "graph", d.get("graph"), "distinct late", d.get("distinct_late"), "early", d.get("distinct_early"), "| %d parameters vary" % len(v))
for k in list(v)[:60]:
    print("     %-80s %d" % (k, v[k]))
PY
}
echo "== replays of one step, two ranks on one device" | tee $OUT/summary.txt
run bf16_graph --precision bf16
run bf16_eager --precision bf16 --no-graph
run f32_graph
echo "== done" | tee -a $OUT/summary.txt

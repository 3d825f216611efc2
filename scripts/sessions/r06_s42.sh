#!/bin/bash
# Round 6 session 42: bf16, two ranks on one device, replays of one eager step WITHOUT the exchange: does the forward vary, and which
# parameters' own gradients do?
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s42}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 CL3D_BENCH_ONE_DEVICE=1 CL3D_DP_NOEXCHANGE=1
run() { # name, args
  local name=$1; shift
  timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --gpus 2 --warmup 1 --gemm-plans model --weight-grads joined --repeat-check 24 --no-graph --dump-forward x "$@" 2>$OUT/$name.err | grep '^{' | tail -1 > $OUT/$name.json
  echo "-- $name" | tee -a $OUT/summary.txt
  grep varying_forward $OUT/$name.err | cut -c1-600 | tee -a $OUT/summary.txt
  python - $OUT/$name.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
except Exception as e:
    print("no line:", e); sys.exit(0)
v = d.get("varying_parameters", {})
print("distinct late", d.get("distinct_late"), "early", d.get("distinct_early"), "| %d parameters vary" % len(v))
ks = list(v)
for k in ks[-40:]:
    print("     %-80s %d" % (k, v[k]))
PY
}
echo "== replays of one eager step without the exchange, two ranks on one device" | tee $OUT/summary.txt
run bf16 --precision bf16
run f32
echo "== done" | tee -a $OUT/summary.txt

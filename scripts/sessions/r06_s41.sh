#!/bin/bash
# Round 6 session 41: which module's OUTPUT first differs between runs of the bf16 step under two ranks on one device (session 40: every
# replay distinct, f32 exact)?  --dump-forward: bit checksums of every module's output in the first eager step.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s41}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 CL3D_BENCH_ONE_DEVICE=1
for i in 1 2 3 4; do
  timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --gpus 2 --warmup 1 --steps 1 --gemm-plans model --weight-grads joined --no-graph --dump-forward $OUT/fwd_$i.json >/dev/null 2>>$OUT/err.log
done
python - $OUT <<'PY' | tee $OUT/summary.txt
import json, sys, os
out = sys.argv[1]
runs = []
for i in range(1, 5):
    try:
        runs.append(json.load(open(os.path.join(out, "fwd_%d.json" % i))))
    except Exception as e:
        print("run", i, "unreadable:", e)
print("== first differing module outputs, runs against run 1 (%d module outputs per run)" % (len(runs[0]) if runs else 0))
for j in range(1, len(runs)):
    a, b = runs[0], runs[j]
    diffs = [k for k in range(min(len(a), len(b))) if a[k] != b[k]]
    print("-- run %d: %d of %d differ; first:" % (j + 1, len(diffs), min(len(a), len(b))))
    for k in diffs[:6]:
        print("    #%d %s" % (k, a[k][0]))
PY
rm -f $OUT/fwd_[2-4].json
echo "== done" | tee -a $OUT/summary.txt

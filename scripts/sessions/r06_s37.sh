#!/bin/bash
# Round 6 session 37: session 36's two-rank checksum read grad_l2 9.2927 (joined) against 8.8440 (deferred) where one process reads
# the same bits either way -- which parameters differ, and is either mode stable from run to run?
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s37}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== two ranks on one device (gloo), config 2 bf16, model plans: per-parameter gradient norms" | tee $OUT/summary.txt
for w in joined deferred joined deferred; do
  i=$((i+1))
  CL3D_BENCH_ONE_DEVICE=1 timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --gpus 2 --steps 20 --checksums --gemm-plans model --weight-grads $w 2>>$OUT/err.log | grep '^{' | tail -1 > $OUT/two_${i}_$w.json
done
for w in joined deferred; do
  timeout 400 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 20 --checksums --gemm-plans model --weight-grads $w 2>>$OUT/err.log | grep '^{' | tail -1 > $OUT/one_$w.json
done
python - $OUT <<'PY' | tee -a $OUT/summary.txt
import json, sys, glob, os
out = sys.argv[1]
runs = {}
for f in sorted(glob.glob(os.path.join(out, "two_[0-9]_*.json")) + glob.glob(os.path.join(out, "one_*.json"))):
    try:
        d = json.loads(open(f).read())
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    runs[os.path.basename(f)] = d
    print(os.path.basename(f), d.get("ms_per_step"), d.get("grad_l2"), d.get("param_l2"))
def diff(a, b):
    A, B = dict(runs[a]["grad_l2_by_param"]), dict(runs[b]["grad_l2_by_param"])
    rows = [(abs(A[k] - B[k]) / (abs(A[k]) + 1e-30), k, A[k], B[k]) for k in A if k in B]
    rows.sort(reverse=True)
    nz = [r for r in rows if r[0] > 0]
    print("-- %s vs %s: %d of %d parameters differ in gradient norm" % (a, b, len(nz), len(rows)))
    for r in nz[:12]:
        print("   %.3e  %-50s %.9g %.9g" % r)
names = sorted(runs)
two = [n for n in names if n.startswith("two_")]
for i in range(len(two)):
    for j in range(i + 1, len(two)):
        diff(two[i], two[j])
if "one_joined.json" in runs and "one_deferred.json" in runs:
    diff("one_joined.json", "one_deferred.json")
PY
echo "== done" | tee -a $OUT/summary.txt

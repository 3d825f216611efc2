#!/bin/bash
OUT=gpurun_out/r04u
mkdir -p $OUT
export TMPDIR=/tmp
echo "== sphere crop" | tee $OUT/summary.txt
timeout 600 python -m pytest tests/test_sphere_crop.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $OUT/summary.txt
echo "== run-to-run determinism of the backbone step across processes" | tee -a $OUT/summary.txt
i=0
for cfg in "--gpus 1 --no-graph" "--gpus 1" "--gpus 2 --no-graph" "--gpus 2 --no-overlap" "--gpus 2 --overlap --overlap-forks none"; do
  i=$((i+1))
  for r in 1 2 3; do
    extra=""
    case "$cfg" in *no-graph*) extra="--dump-forward /tmp/f_${i}_$r.json";; esac
    timeout 300 python scripts/bench_backbone.py $cfg --config modelnet_small --warmup 1 --head --dump-grads /tmp/g_${i}_$r.pt $extra > $OUT/run_${i}_$r.log 2>&1 || echo "cfg $i run $r failed" | tee -a $OUT/summary.txt
  done
done
python - <<'P' 2>&1 | tee -a $OUT/summary.txt
import json, os, torch
names = ["1 rank eager", "1 rank graph", "2 ranks eager", "2 ranks one graph", "2 ranks two graphs no forks"]
for i, nm in enumerate(names, 1):
    try:
        d = [torch.load(f"/tmp/g_{i}_{r}.pt") for r in (1, 2, 3)]
    except Exception as e:
        print(nm, "missing", e); continue
    for r in (1, 2):
        bad = [k for k in d[0] if not torch.equal(d[0][k], d[r][k])]
        print(nm, f"run 1 vs run {r+1}: differing gradient tensors {len(bad)} of {len(d[0])}", bad[:4])
    if os.path.exists(f"/tmp/f_{i}_1.json"):
        f = [json.load(open(f"/tmp/f_{i}_{r}.json")) for r in (1, 2, 3)]
        for r in (1, 2):
            first = next(((k, a[0]) for k, (a, b) in enumerate(zip(f[0], f[r])) if a != b), None)
            print(nm, f"forward run 1 vs run {r+1}: {len(f[0])} module outputs, first differing:", first)
P
echo "== done" | tee -a $OUT/summary.txt

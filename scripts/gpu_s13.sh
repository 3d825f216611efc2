#!/bin/bash
OUT=gpurun_out/r04t
mkdir -p $OUT
export TMPDIR=/tmp
echo "== two-graph step: forks in A (x4) against no forks (x2)" | tee $OUT/summary.txt
for tag in none1 a1 a2 none2 a3 a4; do
  mode=${tag%?}
  timeout 300 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --warmup 1 --head --overlap --overlap-forks $mode --dump-grads /tmp/g_$tag.pt > $OUT/run_$tag.log 2>&1
  echo "$tag rc=$?" | tee -a $OUT/summary.txt
done
python - <<'P' 2>&1 | tee -a $OUT/summary.txt
import torch
d = {t: torch.load(f"/tmp/g_{t}.pt") for t in ("none1", "none2", "a1", "a2", "a3", "a4")}
ref = d["none1"]
for t, g in d.items():
    bad = []
    for k, v in ref.items():
        if not torch.equal(g[k], v):
            bad.append((float((g[k] - v).abs().max() / (v.abs().max() + 1e-30)), k))
    print(t, "differing tensors:", len(bad), "of", len(ref), [(f"{r:.2e}", k) for r, k in sorted(bad, reverse=True)[:6]], [k for _, k in bad][:40])
P
echo "== the whole GPU suite in one process" | tee -a $OUT/summary.txt
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/suite.log 2>&1
echo "suite rc=$?" | tee -a $OUT/summary.txt
grep -n -m1 -A30 "Fatal Python error" $OUT/suite.log | cut -c1-200 | tee -a $OUT/summary.txt
tail -5 $OUT/suite.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

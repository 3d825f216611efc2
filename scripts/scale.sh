#!/bin/bash
# One command for the day an 8-GPU node is there: N = 1, 2, 4, 8 for the LocalAggregation bench (bench.py, the
# driver's contract) and for the backbone steps of configs 4 and 5 (per-GPU shards, overlapped gradient exchange).
#   bash scripts/scale.sh [outdir]
OUT=${1:-gpurun_out/scale}
mkdir -p $OUT
for n in 1 2 4 8; do
  if [ $n -eq 1 ]; then
    python bench.py --gpus 1 --no-cpu-baseline --no-kernel-roofline --backbone on | tee $OUT/la_n$n.json
    for c in partnet_adaptive s3dis_pospool_deep modelnet_pointwisemlp; do
      python scripts/bench_backbone.py --config $c | tail -1 | tee $OUT/${c}_n$n.json
    done
  else
    # (both scripts re-launch themselves as N ranks when WORLD_SIZE is unset: closerlook3d_amd.dp.torchrun_command,
    #  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P <script> ...`)
    python bench.py --gpus $n --no-cpu-baseline --no-kernel-roofline --backbone on | grep '^{' | tee $OUT/la_n$n.json
    for c in partnet_adaptive s3dis_pospool_deep modelnet_pointwisemlp; do
      python scripts/bench_backbone.py --gpus $n --config $c | grep '^{' | tee $OUT/${c}_n$n.json
    done
  fi
done
python - <<PY
import glob, json, os
rows = {}
for p in sorted(glob.glob("$OUT/*_n*.json")):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        continue
    name, n = os.path.basename(p).rsplit("_n", 1)
    rows.setdefault(name, {})[int(n.split(".")[0])] = d.get("value") or d.get("input_points_per_s")
    bb = d.get("backbone_step")  # bench.py's second figure: the BASELINE backbone step incl. the gradient all-reduce
    if isinstance(bb, dict) and "input_points_per_s" in bb:
        rows.setdefault(name + ".backbone_step(" + bb["config"] + ")", {})[int(n.split(".")[0])] = bb["input_points_per_s"]
        if "distinct_devices" in d.get("config", {}):
            print(os.path.basename(p), "ranks on", d["config"]["distinct_devices"], "distinct devices, backend", d["config"].get("backend"),
                  "| all-reduce", bb.get("allreduce_bytes"), "B in", bb.get("allreduce_ms"), "ms")
for name, r in rows.items():
    base = r.get(1)
    print(name, {n: (round(v / 1e6, 2), round(v / base / n, 3) if base else None) for n, v in sorted(r.items())},
          "(M points/s, efficiency vs N = 1)")
PY

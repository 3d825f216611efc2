#!/bin/bash
OUT=gpurun_out/r05i
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1 ) 2> $OUT/pytest_time.txt
echo "pytest rc=$?" | tee $OUT/summary.txt; tail -12 $OUT/pytest_gpu.log | cut -c1-300 | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_time.txt | tee -a $OUT/summary.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err ) 2>> $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d = json.load(open("$OUT/bench_driver_flags.json")); r = d["roofline"]
print("ms_per_step", d["ms_per_step"], d["config"]["launch"], "| top", r["kernel"], r["us"], "frac", r["frac"], "| boundary", r["boundary"]["ball_query_group"]["frac"], "| achieved_step", r["achieved_step"]["frac"])
print("backbone_step", d.get("backbone_step"))
print("contraction", {k: r["contraction"][k] for k in ("us", "fwd_us", "bwd_both_us", "separate_products_us", "frac", "hbm_frac")})
PY
for v in "" "CL3D_CSR_FIRST=1"; do
  for prec in f32 bf16; do
    env $v timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision $prec 2>/dev/null | tail -1 | cut -c1-230 | sed "s/^/[$v] /" | tee -a $OUT/summary.txt
  done
done

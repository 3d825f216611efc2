#!/bin/bash
# Round-3 session A: parity of the new kernels, the bench line, A/B of the new paths.  bash scripts/gpu_r03a.sh [tag]
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== smoke" | tee $OUT/summary.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
echo "== ball query + operator parity first (fast fail)" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_native_gpu.py tests/test_bq_paths_gpu.py tests/test_ref_pin_gpu.py -m gpu -q -x --timeout=600 -p no:cacheprovider > $OUT/pytest_bq.log 2>&1
echo "pytest bq rc=$?" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest_bq.log | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider --deselect tests/test_native_gpu.py --deselect tests/test_bq_paths_gpu.py --deselect tests/test_ref_pin_gpu.py > $OUT/pytest_rest.log 2>&1
echo "pytest rest rc=$?" | tee -a $OUT/summary.txt; tail -8 $OUT/pytest_rest.log | tee -a $OUT/summary.txt
echo "== bench (driver's command)" | tee -a $OUT/summary.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
try:
    d = json.load(open("$OUT/bench.json"))
    r = d["roofline"]
    print("ms_per_step", d["ms_per_step"], "value", d["value"])
    print("top", r.get("kernel"), r.get("us"), r.get("frac"), "boundary", r["boundary"]["ball_query_group"])
    for k in r["step"]["kernels"]:
        print("  %-40s %7.2f us (%.2f-%.2f)" % (k["entry"], k["us"], k["us_min"], k["us_max"]))
    for k, v in r["boundary"]["per_kernel"].items():
        print("  boundary %-40s %.5f ms (%.5f-%.5f) frac %.4f" % (k, v["ms"], v["ms_min"], v["ms_max"], v["frac"]))
    print("cpu", d.get("cpu_baseline", {}).get("all_cores"), d.get("cpu_baseline", {}).get("one_thread"))
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/bench.err").read()[-2000:])
PY
echo "== A/B: ball query paths" | tee -a $OUT/summary.txt
for p in tile cells; do CL3D_BQ_PATH=$p timeout 120 python scripts/bench_bq.py | tee -a $OUT/summary.txt; done
CL3D_BQ_PATH=tile timeout 120 python scripts/bench_bq.py --n 1024 --k 32 | tee -a $OUT/summary.txt
CL3D_BQ_PATH=cells timeout 120 python scripts/bench_bq.py --n 1024 --k 32 | tee -a $OUT/summary.txt
CL3D_BQ_PATH=tile timeout 120 python scripts/bench_bq.py --mult 4.0 | tee -a $OUT/summary.txt
CL3D_BQ_PATH=cells timeout 120 python scripts/bench_bq.py --mult 4.0 | tee -a $OUT/summary.txt
echo "== A/B: step variants (no cpu baseline, no boundary)" | tee -a $OUT/summary.txt
for v in "" "CL3D_PW_QPG=1" "CL3D_PW_SB=6" "CL3D_BQ_PATH=cells"; do
  echo "-- $v" | tee -a $OUT/summary.txt
  env $v timeout 300 python bench.py --no-cpu-baseline --bursts 4 2>/dev/null > $OUT/bench_v.json
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    d = json.load(open("$OUT/bench_v.json")); r = d["roofline"]
    print("ms_per_step", d["ms_per_step"], " ".join("%s=%.1f" % (k["entry"].replace("cl3d_", ""), k["us"]) for k in r["step"]["kernels"][:8]))
except Exception as e:
    print("failed", e)
PY
done
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
OUT=gpurun_out/r04d
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== suites: ball query (flat window), CSR entries + summary + support pass, operators" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests/test_native_gpu.py tests/test_bq_paths_gpu.py tests/test_ref_pin_gpu.py tests/test_fullsize_gpu.py tests/test_pwmlp_summary_gpu.py tests/test_operators_gpu.py tests/test_capture_gpu.py tests/test_scene_size_gpu.py tests/test_bottleneck_gpu.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -12 | tee -a $OUT/summary.txt
echo "== variants" | tee -a $OUT/summary.txt
timeout 1500 python scripts/micro/bq_variants.py --run --step 2>&1 | tee $OUT/variants.jsonl | tee -a $OUT/summary.txt
for p in tile tile1 cells; do CL3D_BQ_PATH=$p timeout 120 python scripts/bench_bq.py --mult 4.0 | tee -a $OUT/summary.txt; done
for p in tile tile1; do CL3D_BQ_PATH=$p timeout 120 python scripts/bench_bq.py --n 1024 | tee -a $OUT/summary.txt; done
echo "== step with the shipped build: per-entry table" | tee -a $OUT/summary.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null > $OUT/bench.json
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open("$OUT/bench.json"))
print("ms_per_step", d["ms_per_step"])
st=d["roofline"].get("step")
rows = st if isinstance(st, list) else (st.get("entries") or st.get("rows") or st)
print(json.dumps(rows)[:3000])
PY
echo "== timeline" | tee -a $OUT/summary.txt
(cd /tmp && rm -rf /tmp/tl && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $R/bench.py --steps 12 --warmup 4 --precondition 0 --no-cpu-baseline --no-kernel-roofline --no-step-table > /dev/null 2>&1)
python scripts/step_timeline.py "/tmp/tl/**/tl_kernel_trace.csv" | tee $OUT/step_timeline.txt | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# round-4 session 1: new LDS-resident ball query (packed distances, window in registers) vs the round-3 kernel; XCD-aware summary
OUT=gpurun_out/r04a
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bit-exact suites" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_native_gpu.py tests/test_bq_paths_gpu.py tests/test_d2_form.py tests/test_ref_pin_gpu.py tests/test_fullsize_gpu.py tests/test_pwmlp_summary_gpu.py tests/test_scene_size_gpu.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -15 | tee -a $OUT/summary.txt
echo "== ball query alone" | tee -a $OUT/summary.txt
for p in tile tile1; do CL3D_BQ_PATH=$p timeout 120 python scripts/bench_bq.py | tee -a $OUT/summary.txt; done
for p in tile tile1; do CL3D_BQ_PATH=$p timeout 120 python scripts/bench_bq.py --n 1024 | tee -a $OUT/summary.txt; done
for p in tile tile1 cells; do CL3D_BQ_PATH=$p timeout 120 python scripts/bench_bq.py --mult 4.0 | tee -a $OUT/summary.txt; done
for p in tile tile1; do CL3D_BQ_PATH=$p timeout 120 python scripts/bench_bq.py --n 4096 --m 1024 | tee -a $OUT/summary.txt; done
echo "== step" | tee -a $OUT/summary.txt
for v in "CL3D_BQ_PATH=tile" "CL3D_BQ_PATH=tile1" "CL3D_CSR_SCAN=fused"; do
  env $v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null > $OUT/bench_$v.json
  python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open("$OUT/bench_$v.json"))
print("$v", "ms_per_step", d["ms_per_step"])
for r in d["roofline"].get("step", []):
    print("   ", r.get("entry"), r.get("us"))
PY
done
echo "== timeline" | tee -a $OUT/summary.txt
R=$GRAFT_REPO_ROOT
(cd /tmp && rm -rf /tmp/tl && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $R/bench.py --steps 12 --warmup 4 --precondition 0 --no-cpu-baseline --no-kernel-roofline --no-step-table > /dev/null 2>&1)
python scripts/step_timeline.py "/tmp/tl/**/tl_kernel_trace.csv" | tee $OUT/step_timeline.txt | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# rotating-pair slot walk of the query-major passes (CL3D_PW_PIPE=1): parity, then A/B
TAG=${1:-r03m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity (slot table + rotating-pair walk on)" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests/test_operators_gpu.py tests/test_fp64_anchor_gpu.py tests/test_fullsize_gpu.py tests/test_bottleneck_gpu.py tests/test_pwmlp_summary_gpu.py tests/test_config2_fullsize_gpu.py tests/test_scene_size_gpu.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest.log | tee -a $OUT/summary.txt
for v in "" "CL3D_PW_PIPE=0" "" "CL3D_PW_PIPE=0"; do
  echo "-- $v" | tee -a $OUT/summary.txt
  env $v timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>$OUT/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" | tee -a $OUT/summary.txt
done
echo "== step table" | tee -a $OUT/summary.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null > $OUT/bench.json
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"])
for k in d["roofline"]["step"]["kernels"]:
    print(f"  {k['entry']:40s} {k['us']:7.2f} us  (min {k['us_min']}, max {k['us_max']})  hbm_frac {k['hbm_frac']}")
PY
for v in ""; do
for prec in f32 bf16; do
  env $v timeout 300 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision $prec 2>/dev/null | tail -1 | cut -c150-260 | tee -a $OUT/summary.txt
done; done
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# support-major pass on the CSR summary: parity, then A/B of the replayed step and the per-entry table
TAG=${1:-r03j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== summary tests + PointWiseMLP parity" | tee $OUT/summary.txt
[ -n "$SKIP_TESTS" ] || timeout 1500 python -m pytest tests/test_pwmlp_summary_gpu.py tests/test_operators_gpu.py tests/test_fp64_anchor_gpu.py tests/test_bottleneck_gpu.py tests/test_fullsize_gpu.py tests/test_dp_gpu.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -6 $OUT/pytest.log | tee -a $OUT/summary.txt
for v in "" "CL3D_PW_SUMMARY=0" "CL3D_PW_SB=4" ""; do
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
echo "== timeline" | tee -a $OUT/summary.txt
R=$(pwd)
(cd /tmp && rm -rf /tmp/tl && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $R/bench.py --steps 12 --warmup 4 --precondition 50 --no-cpu-baseline --no-kernel-roofline --no-step-table > /dev/null 2>&1)
python scripts/step_timeline.py "/tmp/tl/**/tl_kernel_trace.csv" | tee $OUT/step_timeline.txt | tee -a $OUT/summary.txt
echo "== backbone config 2" | tee -a $OUT/summary.txt
for prec in f32 bf16; do
  timeout 300 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision $prec 2>/dev/null | tail -1 | cut -c150-260 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

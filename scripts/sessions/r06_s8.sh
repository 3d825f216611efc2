#!/bin/bash
# Round 6 session 8: the BatchNorm + ReLU output transform inside the reduce pass calls: tests, eager steps
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s8
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest" | tee $OUT/summary.txt
timeout 2400 python -m pytest tests/test_pass_calls_gpu.py tests/test_abi_host_gpu.py tests/test_operators_gpu.py tests/test_fp64_anchor_gpu.py tests/test_capture_gpu.py tests/test_bottleneck_gpu.py -q -m gpu --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -8 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== eager steps (bench.py --no-graph)" | tee -a $OUT/summary.txt
for i in 1 2 3; do for op in pointwisemlp pospool adaptive_weight pseudo_grid; do
  timeout 600 python bench.py --operator $op --no-graph --no-cpu-baseline --no-kernel-roofline --backbone off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$op eager', d['ms_per_step'], d['value'])" | tee -a $OUT/summary.txt
done; done
timeout 300 python scripts/micro/eager_host.py 2>/dev/null | head -6 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

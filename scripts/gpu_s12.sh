#!/bin/bash
OUT=gpurun_out/r04s
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pass calls: parity with the kernel-by-kernel path, repeatability; the suites that touch the operator" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_pass_calls_gpu.py tests/test_operators_gpu.py tests/test_capture_gpu.py tests/test_fp64_anchor_gpu.py tests/test_fullsize_gpu.py tests/test_sphere_crop.py tests/test_dp_gpu.py -m gpu -q --timeout=900 -p no:cacheprovider 2>&1 | tail -12 | tee -a $OUT/summary.txt
echo "== eager bench (pass calls) / graph" | tee -a $OUT/summary.txt
for flags in "--no-graph" ""; do
  timeout 300 python bench.py $flags --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags [$flags]', 'ms_per_step', d['ms_per_step'], d['config']['launch'])" | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

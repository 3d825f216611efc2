#!/bin/bash
OUT=gpurun_out/r04be
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mfma_gemm_gpu.py tests/test_bottleneck_gpu.py tests/test_operators_gpu.py tests/test_config2_fullsize_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 | cut -c1-250 | tee $OUT/summary.txt
for i in 1 2; do
  timeout 300 python bench.py --precision bf16 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 ms_per_step', d['ms_per_step'])" | tee -a $OUT/summary.txt
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f32 ms_per_step', d['ms_per_step'])" | tee -a $OUT/summary.txt
done
timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | tail -1 | cut -c1-300 | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_point_gemm.py --sweep 2>/dev/null | cut -c1-400 | tee -a $OUT/summary.txt

#!/bin/bash
OUT=gpurun_out/r04c
mkdir -p $OUT
export TMPDIR=/tmp
echo "== bit-exact suites (default build: window in registers, lock-step ranking)" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_native_gpu.py tests/test_bq_paths_gpu.py tests/test_ref_pin_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -5 | tee -a $OUT/summary.txt
echo "== variants" | tee -a $OUT/summary.txt
timeout 1500 python scripts/micro/bq_variants.py --run --step 2>&1 | tee $OUT/variants.jsonl | tee -a $OUT/summary.txt
CL3D_BQ_PATH=tile1 python scripts/bench_bq.py | tee -a $OUT/summary.txt
CL3D_BQ_PATH=tile1 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | cut -c1-200 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

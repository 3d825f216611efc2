#!/bin/bash
OUT=gpurun_out/r04s
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pass calls" | tee $OUT/summary.txt
for f in tests/test_pass_calls_gpu.py tests/test_operators_gpu.py tests/test_capture_gpu.py tests/test_fp64_anchor_gpu.py tests/test_fullsize_gpu.py tests/test_dp_gpu.py; do
  timeout 900 python -X faulthandler -m pytest $f -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/$(basename $f .py).log 2>&1
  echo "$f rc=$?" | tee -a $OUT/summary.txt
  grep -n -m1 -A25 "Fatal Python error" $OUT/$(basename $f .py).log | head -40 | tee -a $OUT/summary.txt
  tail -3 $OUT/$(basename $f .py).log | cut -c1-300 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

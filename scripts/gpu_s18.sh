#!/bin/bash
OUT=gpurun_out/r04aa
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_pass_calls_gpu.py tests/test_operators_gpu.py tests/test_capture_gpu.py tests/test_fp64_anchor_gpu.py tests/test_bottleneck_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee -a $OUT/summary.txt
run() {  # $1 = QUERY_IN_LINE, rest = bench flags
  q=$1; shift
  timeout 300 python -c "
import sys, runpy
import closerlook3d_amd.fused as f
f.QUERY_IN_LINE = $q
sys.argv = ['bench.py'] + '$*'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in_line=$q [$*]', 'ms_per_step', d['ms_per_step'], d['config']['launch'])" | tee -a $OUT/summary.txt
}
echo "== graph: query in line (product forked) vs query forked" | tee -a $OUT/summary.txt
for rep in 1 2; do
  run True --no-cpu-baseline --no-kernel-roofline
  run False --no-cpu-baseline --no-kernel-roofline
done
echo "== eager (pass calls: query on the caller's stream, product and data gradient forked)" | tee -a $OUT/summary.txt
run True --no-graph --no-cpu-baseline --no-kernel-roofline
run True --no-graph --no-cpu-baseline --no-kernel-roofline
echo "== done" | tee -a $OUT/summary.txt

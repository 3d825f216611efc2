#!/bin/bash
OUT=gpurun_out/r04ab
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # $1 = LEAF_FORKS_DATA_GRAD, rest = bench flags
  q=$1; shift
  timeout 300 python -c "
import sys, runpy
import closerlook3d_amd.fused as f
f.LEAF_FORKS_DATA_GRAD = $q
sys.argv = ['bench.py'] + '$*'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('leaf_forks_data_grad=$q [$*]', 'ms_per_step', d['ms_per_step'], d['config']['launch'])" | tee -a $OUT/summary.txt
}
echo "== graph: a leaf input's data gradient forks (weight gradient on the caller's stream) vs the weight gradient forks" | tee $OUT/summary.txt
for rep in 1 2; do
  run True --no-cpu-baseline --no-kernel-roofline
  run False --no-cpu-baseline --no-kernel-roofline
done
timeout 600 python -m pytest tests/test_operators_gpu.py tests/test_capture_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

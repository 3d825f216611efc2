#!/bin/bash
TAG=${1:-r03i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== dp + prefetch tests" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests/test_dp_gpu.py tests/test_operators_gpu.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest.log | tee -a $OUT/summary.txt
for v in "" "CL3D_BQ_PRIORITY=1"; do
  for i in 1 2; do
  echo "-- $v" | tee -a $OUT/summary.txt
  env $v timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" | tee -a $OUT/summary.txt
  done
done
echo "== backbone config 2 (subsampling chain prefetched)" | tee -a $OUT/summary.txt
for v in "" "CL3D_PREFETCH=0" "CL3D_BQ_PRIORITY=1"; do
  for prec in f32 bf16; do
    echo "-- $v $prec" | tee -a $OUT/summary.txt
    env $v timeout 300 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision $prec 2>/dev/null | tail -1 | cut -c150-260 | tee -a $OUT/summary.txt
  done
done
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# Round-3 session E: sphere-crop kernels, fused bottleneck (operand prologue), ball-query output tweak
TAG=${1:-r03e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_sphere_crop.py tests/test_planner.py tests/test_bottleneck_gpu.py tests/test_native_gpu.py tests/test_operators_gpu.py tests/test_mfma_gemm_gpu.py tests/test_config2_fullsize_gpu.py tests/test_fp64_anchor_gpu.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -15 $OUT/pytest.log | tee -a $OUT/summary.txt
echo "== sphere crop bench" | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_sphere_crop.py 2>&1 | tail -2 | tee -a $OUT/summary.txt
echo "== backbone config 2: fused bottleneck on / off, f32 and bf16" | tee -a $OUT/summary.txt
for v in "" "CL3D_FUSE_BOTTLENECK=0"; do
  for prec in f32 bf16; do
    echo "-- $v $prec" | tee -a $OUT/summary.txt
    env $v timeout 300 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision $prec 2>/dev/null | tail -1 | cut -c1-400 | tee -a $OUT/summary.txt
  done
done
echo "== bench" | tee -a $OUT/summary.txt
timeout 300 python bench.py --no-cpu-baseline --bursts 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms_per_step', d['ms_per_step'], 'boundary', r['boundary']['ball_query_group']['frac'], r['boundary']['per_kernel']['ball_query']['ms'])" | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

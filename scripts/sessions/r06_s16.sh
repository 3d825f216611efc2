#!/bin/bash
# Round 6 session 16: GEMM plans by measurement (cl3d_gemm_autotune) against the launch-time model, replayed backbones.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s16
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_mfma_gemm_gpu.py tests/test_bottleneck_gpu.py tests/test_config2_fullsize_gpu.py -q -m gpu --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d.get('ms_per_step'), d.get('gemm_plans_measured'), d.get('gemm_plans_changed'))" "$1"; }
for cfg in "modelnet_pointwisemlp --precision bf16" "modelnet_pointwisemlp --precision f32" "s3dis_pseudogrid" "partnet_adaptive" "s3dis_pospool_deep"; do
  echo "== backbone $cfg" | tee -a $OUT/summary.txt
  for i in 1 2; do
    timeout 400 python scripts/bench_backbone.py --config $cfg --gemm-plans model 2>$OUT/err.log | line model | tee -a $OUT/summary.txt
    timeout 400 python scripts/bench_backbone.py --config $cfg --gemm-plans measured 2>$OUT/err.log | line measured | tee -a $OUT/summary.txt
  done
done
tail -5 $OUT/err.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

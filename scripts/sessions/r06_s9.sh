#!/bin/bash
# Round 6 session 9: grid subsampling with the radix sort in LDS (tests, timing against the bitonic variant)
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s9
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
echo "== pytest" | tee $OUT/summary.txt
timeout 2400 python -m pytest tests/test_native_gpu.py tests/test_ref_pin_gpu.py tests/test_scene_size_gpu.py tests/test_operators_gpu.py tests/test_mfma_gemm_gpu.py tests/test_capture_gpu.py -q -m gpu --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -6 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== masked_grid_subsampling: radix sort on the cell bits (shipped) / bitonic network (variant)" | tee -a $OUT/summary.txt
for i in 1 2; do
  timeout 300 python scripts/micro/bench_grid_subsample.py | tee -a $OUT/summary.txt
  CL3D_LIB=$R/scripts/micro/var/libcl3d_sub_bitonic.so timeout 300 python scripts/micro/bench_grid_subsample.py | tee -a $OUT/summary.txt
done
echo "== config 2 backbone" | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | grep '^{' | tail -1 | cut -c1-260 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

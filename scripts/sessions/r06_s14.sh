#!/bin/bash
# Round 6 session 14: the GEMM's tiles handed to the XCDs so that the tiles sharing the large operand strip run on ONE
# XCD back to back (shipped) against the plain order (scripts/micro/var/libcl3d_gemm_plain_order.so), alternating runs.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s14
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
V=$R/scripts/micro/var/libcl3d_gemm_plain_order.so
echo "== pytest" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_mfma_gemm_gpu.py tests/test_bottleneck_gpu.py tests/test_config2_fullsize_gpu.py -q -m gpu --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d.get('ms_per_step'), d.get('value'))" "$1"; }
echo "== 1x1 convolutions of config 2 (us: fwd, d x, d W), xcd order" | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_point_gemm.py --convs --reps 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['conv'], 'f32', d['f32'], 'bf16', d['bf16'])" | tee -a $OUT/summary.txt
echo "== the same, plain order" | tee -a $OUT/summary.txt
CL3D_LIB=$V timeout 300 python scripts/bench_point_gemm.py --convs --reps 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['conv'], 'f32', d['f32'], 'bf16', d['bf16'])" | tee -a $OUT/summary.txt
for cfg in "modelnet_pointwisemlp --precision bf16" "modelnet_pointwisemlp --precision f32" "s3dis_pseudogrid" "partnet_adaptive" "s3dis_pospool_deep"; do
  echo "== backbone $cfg" | tee -a $OUT/summary.txt
  for i in 1 2; do
    timeout 400 python scripts/bench_backbone.py --config $cfg 2>/dev/null | line xcd | tee -a $OUT/summary.txt
    CL3D_LIB=$V timeout 400 python scripts/bench_backbone.py --config $cfg 2>/dev/null | line plain | tee -a $OUT/summary.txt
  done
done
echo "== headline" | tee -a $OUT/summary.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | line xcd | tee -a $OUT/summary.txt
CL3D_LIB=$V timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | line plain | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# Round 6 session 13: BatchNorm statistics finished by the last-arriving workgroup (no finalize launch); the GEMM's
# in-launch slice sum with an explicit wait for the slice stores in front of its ticket.  A/B against the variant library
# that keeps the finalize launch (scripts/micro/var/libcl3d_bn_finalize_launch.so), alternating runs.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s13
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
V=$R/scripts/micro/var/libcl3d_bn_finalize_launch.so
echo "== pytest" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_operators_gpu.py tests/test_bottleneck_gpu.py tests/test_mfma_gemm_gpu.py tests/test_pass_calls_gpu.py tests/test_fp64_anchor_gpu.py tests/test_capture_gpu.py -q -m gpu --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -6 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d.get('ms_per_step'), d.get('value'))" "$1"; }
echo "== headline (PointWiseMLP; no BatchNorm in it -- the GEMM ticket change only)" | tee -a $OUT/summary.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | line shipped | tee -a $OUT/summary.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | line shipped | tee -a $OUT/summary.txt
echo "== pospool operator step, folded / finalize launch, alternating" | tee -a $OUT/summary.txt
for i in 1 2 3; do
  timeout 300 python bench.py --operator pospool --no-cpu-baseline 2>/dev/null | line folded | tee -a $OUT/summary.txt
  CL3D_LIB=$V timeout 300 python bench.py --operator pospool --no-cpu-baseline 2>/dev/null | line finalize_launch | tee -a $OUT/summary.txt
done
echo "== adaptive_weight operator step" | tee -a $OUT/summary.txt
for i in 1 2; do
  timeout 300 python bench.py --operator adaptive_weight --no-cpu-baseline 2>/dev/null | line folded | tee -a $OUT/summary.txt
  CL3D_LIB=$V timeout 300 python bench.py --operator adaptive_weight --no-cpu-baseline 2>/dev/null | line finalize_launch | tee -a $OUT/summary.txt
done
echo "== config 2 backbone bf16" | tee -a $OUT/summary.txt
for i in 1 2; do
  timeout 400 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | line folded | tee -a $OUT/summary.txt
  CL3D_LIB=$V timeout 400 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | line finalize_launch | tee -a $OUT/summary.txt
done
echo "== config 5 backbone (s3dis_pospool_deep)" | tee -a $OUT/summary.txt
timeout 400 python scripts/bench_backbone.py --config s3dis_pospool_deep 2>/dev/null | line folded | tee -a $OUT/summary.txt
CL3D_LIB=$V timeout 400 python scripts/bench_backbone.py --config s3dis_pospool_deep 2>/dev/null | line finalize_launch | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

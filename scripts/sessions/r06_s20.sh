#!/bin/bash
# Round 6 session 20: the new lane maps (fewer, wider row pieces per wave-load) against rounds 1-5's (variant lane_rule_r5):
# tests, the four operator steps, the five backbones, alternating runs.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s20
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
V=$R/scripts/micro/var/libcl3d_lane_rule_r5.so
echo "== pytest" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_operators_gpu.py tests/test_bottleneck_gpu.py tests/test_pass_calls_gpu.py tests/test_fp64_anchor_gpu.py tests/test_scene_size_gpu.py tests/test_config2_fullsize_gpu.py -q -m gpu --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d.get('ms_per_step'), d.get('value'))" "$1"; }
for op in pospool adaptive_weight pseudo_grid pointwisemlp; do
  echo "== operator step $op" | tee -a $OUT/summary.txt
  for i in 1 2; do
    timeout 300 python bench.py --operator $op --no-cpu-baseline --backbone off 2>/dev/null | line new | tee -a $OUT/summary.txt
    CL3D_LIB=$V timeout 300 python bench.py --operator $op --no-cpu-baseline --backbone off 2>/dev/null | line r5 | tee -a $OUT/summary.txt
  done
done
for cfg in "modelnet_pointwisemlp --precision bf16" "modelnet_pointwisemlp --precision f32" "s3dis_pseudogrid" "partnet_adaptive" "s3dis_pospool_deep"; do
  echo "== backbone $cfg" | tee -a $OUT/summary.txt
  for i in 1 2; do
    timeout 400 python scripts/bench_backbone.py --config $cfg 2>/dev/null | line new | tee -a $OUT/summary.txt
    CL3D_LIB=$V timeout 400 python scripts/bench_backbone.py --config $cfg 2>/dev/null | line r5 | tee -a $OUT/summary.txt
  done
done
echo "== done" | tee -a $OUT/summary.txt

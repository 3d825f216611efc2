#!/bin/bash
# Round 6 session 21: max pooling with kept support indices and a scattered gradient against the CSR gather form
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s21
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_operators_gpu.py tests/test_bottleneck_gpu.py tests/test_scene_size_gpu.py tests/test_config2_fullsize_gpu.py tests/test_abi_host_gpu.py tests/test_capture_gpu.py -q -m gpu --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -6 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d.get('ms_per_step'), d.get('value'))" "$1"; }
for cfg in "modelnet_pointwisemlp --precision bf16" "s3dis_pseudogrid" "partnet_adaptive" "s3dis_pospool_deep"; do
  echo "== backbone $cfg" | tee -a $OUT/summary.txt
  for i in 1 2; do
    timeout 400 python scripts/bench_backbone.py --config $cfg --maxpool targets 2>/dev/null | line targets | tee -a $OUT/summary.txt
    timeout 400 python scripts/bench_backbone.py --config $cfg --maxpool slots 2>/dev/null | line slots | tee -a $OUT/summary.txt
  done
done
echo "== done" | tee -a $OUT/summary.txt

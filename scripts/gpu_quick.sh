#!/bin/bash
TAG=${1:-r02k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest subset" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_mfma_gemm_gpu.py tests/test_bottleneck_gpu.py tests/test_scene_size_gpu.py tests/test_planner.py tests/test_operators_gpu.py -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; grep -E "passed|failed|^FAILED|Error:" $OUT/pytest.log | tail -8 | tee -a $OUT/summary.txt
for i in 1 2; do
echo "== bench f32 / bf16 (run $i)" | tee -a $OUT/summary.txt
timeout 600 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | cut -c1-200 | tee -a $OUT/summary.txt
CL3D_ASYNC=0 timeout 600 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | cut -c1-200 | sed 's/^/no side streams: /' | tee -a $OUT/summary.txt
timeout 600 python bench.py --no-cpu-baseline --no-kernel-roofline --precision bf16 2>/dev/null | cut -c1-200 | tee -a $OUT/summary.txt
done
echo "== backbone" | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp 2>/dev/null | tail -1 | cut -c1-330 | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | tail -1 | cut -c1-330 | tee -a $OUT/summary.txt
CL3D_BLOCK=modules timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp 2>/dev/null | tail -1 | cut -c1-330 | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_backbone.py --config s3dis_pseudogrid 2>/dev/null | tail -1 | cut -c1-330 | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_backbone.py --config partnet_adaptive 2>/dev/null | tail -1 | cut -c1-330 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

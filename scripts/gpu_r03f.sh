#!/bin/bash
# Round-3 session F: GEMM changes, sphere crop with compaction, overlapped DP exchange, late join
TAG=${1:-r03f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests" | tee $OUT/summary.txt
timeout 1800 python -m pytest tests/test_mfma_gemm_gpu.py tests/test_bottleneck_gpu.py tests/test_operators_gpu.py tests/test_sphere_crop.py tests/test_planner.py tests/test_dp_gpu.py tests/test_fp64_anchor_gpu.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -12 $OUT/pytest.log | tee -a $OUT/summary.txt
echo "== sphere crop bench" | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_sphere_crop.py 2>&1 | tail -1 | tee -a $OUT/summary.txt
echo "== point gemm (metric shape)" | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_point_gemm.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in ('mfma_f32','mfma_bf16','library_f32'):
    if k in d: print(k, {a: round(b,1) for a,b in d[k].items() if a.endswith('_us')})
" | tee -a $OUT/summary.txt
for sp in 128 256 384; do
  echo "-- CL3D_GEMM_SPLIT=$sp" | tee -a $OUT/summary.txt
  CL3D_GEMM_SPLIT=$sp timeout 300 python scripts/bench_point_gemm.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bwd_weight f32', round(d['mfma_f32']['bwd_weight_us'],1), 'bf16', round(d['mfma_bf16']['bwd_weight_us'],1))
" | tee -a $OUT/summary.txt
done
echo "== convs" | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_point_gemm.py --convs --reps 20 2>/dev/null > $OUT/convs.jsonl; head -c 1500 $OUT/convs.jsonl | tee -a $OUT/summary.txt
echo "== bench variants" | tee -a $OUT/summary.txt
for v in "" "CL3D_LATE_JOIN=1" "CL3D_PW_QPG=2"; do
  echo "-- $v" | tee -a $OUT/summary.txt
  env $v timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>$OUT/bench_err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['config']['launch'])" | tee -a $OUT/summary.txt
  grep -i "capture failed" $OUT/bench_err.log | tee -a $OUT/summary.txt
done
echo "== backbone config 2" | tee -a $OUT/summary.txt
for prec in f32 bf16; do
  timeout 300 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision $prec 2>/dev/null | tail -1 | cut -c1-330 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

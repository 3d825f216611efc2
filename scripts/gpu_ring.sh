#!/bin/bash
# A/B of the LDS-DMA ring GEMM (CL3D_GEMM_RING=1, default) against the register-staged kernel (=0)
TAG=${1:-ring}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest gemm + bottleneck (ring on)" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_mfma_gemm_gpu.py tests/test_bottleneck_gpu.py -m gpu -q --timeout=600 -p no:cacheprovider -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; grep -E "passed|failed|^FAILED|Error|assert" $OUT/pytest.log | tail -8 | tee -a $OUT/summary.txt
for ring in 1 0; do
echo "== CL3D_GEMM_RING=$ring: point GEMMs [fwd, d features, d weight] us" | tee -a $OUT/summary.txt
CL3D_GEMM_RING=$ring timeout 300 python scripts/bench_point_gemm.py --sweep --reps 30 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    s = d['shape']
    print(s['C'], s['N'], ' | '.join('%s f %.1f d %.1f w %.1f' % (k, d[k]['fwd_us'], d[k]['bwd_data_us'], d[k]['bwd_weight_us']) for k in ('mfma_f32', 'library_f32')))
" | tee -a $OUT/summary.txt
echo "== CL3D_GEMM_RING=$ring: convolutions f32 [fwd, dx, dW]" | tee -a $OUT/summary.txt
CL3D_GEMM_RING=$ring timeout 600 python scripts/bench_point_gemm.py --convs --reps 20 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); c = d['conv']
    print('%4d->%4d N=%4d  f32 %s  lib %s' % (c['C'], c['Co'], c['N'], d['f32'], d['library_f32']))
" | tee -a $OUT/summary.txt
echo "== CL3D_GEMM_RING=$ring: bench + backbone f32" | tee -a $OUT/summary.txt
CL3D_GEMM_RING=$ring timeout 600 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | cut -c1-200 | tee -a $OUT/summary.txt
CL3D_GEMM_RING=$ring timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp 2>/dev/null | tail -1 | cut -c1-300 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

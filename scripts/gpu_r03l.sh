#!/bin/bash
# vectorised transpose: parity, then the operator benches that transpose at their boundary
TAG=${1:-r03l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests/test_transpose_gpu.py tests/test_operators_gpu.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest.log | tee -a $OUT/summary.txt
python - <<'PY' | tee -a $OUT/summary.txt
import torch, time
from closerlook3d_amd import fused
for shape in [(16,64,4096),(16,4096,64),(16,72,4096),(16,4096,72),(1,144,40960)]:
    x=torch.randn(*shape,device='cuda')
    for _ in range(5): fused._transposed(x)
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): fused._transposed(x)
    b.record(); torch.cuda.synchronize()
    us=a.elapsed_time(b)*1e3/50
    print(f"transpose {shape}: {us:.1f} us (incl. launch from Python), {2*x.numel()*4/us/1e6:.2f} TB/s")
PY
for op in pospool adaptive_weight pseudo_grid; do
  timeout 300 python bench.py --operator $op --no-cpu-baseline --no-kernel-roofline --no-step-table 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$op ms_per_step', d['ms_per_step'], 'M points/s', round(d['value']/1e6,1))" | tee -a $OUT/summary.txt
done
for cfg in s3dis_pseudogrid partnet_adaptive s3dis_pospool_deep; do
  timeout 300 python scripts/bench_backbone.py --config $cfg 2>/dev/null | tail -1 | cut -c1-60,150-230 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# The experimental CSR build without its scan launch (csrc/csr.hip, CL3D_CSR_SCAN=fused): parity first, then
# the A/B.  Written at the end of round 3 after the GPU budget was spent -- run this first thing next round:
#   gpurun --timeout 600 -- 'bash scripts/gpu_csr_fused.sh'
export TMPDIR=/tmp
export CL3D_CSR_SCAN=fused
echo "== parity under CL3D_CSR_SCAN=fused (CSR vs numpy, scene sizes, every operator's gradients, the summary)"
timeout 500 python -m pytest tests/test_operators_gpu.py tests/test_scene_size_gpu.py tests/test_pwmlp_summary_gpu.py tests/test_fp64_anchor_gpu.py \
  -m gpu -q -x --timeout=300 -p no:cacheprovider 2>&1 | tail -3
unset CL3D_CSR_SCAN
echo "== replayed steps, alternating (off / fused)"
for op in pointwisemlp pseudo_grid pospool; do
  for v in "" "CL3D_CSR_SCAN=fused" "" "CL3D_CSR_SCAN=fused"; do
    env $v timeout 200 python bench.py --operator $op --no-cpu-baseline --no-kernel-roofline 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$op', '${v:-off}', 'ms_per_step', d['ms_per_step'])"
  done
done

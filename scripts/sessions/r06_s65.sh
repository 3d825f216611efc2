#!/bin/bash
# Round 6 session 65: the TRAIN walk on scalar FMAs (variant train_scalar, -DCL3D_TRAIN_PK=0) against the shipped packed pairs:
# the headline step and its TRAIN entry, alternating runs; config 2 bf16
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s65}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
V=$PWD/scripts/micro/var/libcl3d_train_scalar.so
line() { grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print(sys.argv[1], d.get('ms_per_step'), 'TRAIN us', r.get('us'))" "$1"; }
echo "== headline, shipped / train_scalar, alternating" | tee $OUT/summary.txt
for i in 1 2 3; do
  timeout 300 python bench.py --steps 100 --backbone off --no-cpu-baseline 2>>$OUT/err.log | line shipped | tee -a $OUT/summary.txt
  CL3D_LIB=$V timeout 300 python bench.py --steps 100 --backbone off --no-cpu-baseline 2>>$OUT/err.log | line train_scalar | tee -a $OUT/summary.txt
done
echo "== config 2 bf16 backbone, shipped / train_scalar" | tee -a $OUT/summary.txt
for i in 1 2; do
  timeout 400 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 30 2>>$OUT/err.log | line shipped | tee -a $OUT/summary.txt
  CL3D_LIB=$V timeout 400 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 30 2>>$OUT/err.log | line train_scalar | tee -a $OUT/summary.txt
done
echo "== parity of the variant (operators, anchors, rows)" | tee -a $OUT/summary.txt
CL3D_LIB=$V timeout 900 python -m pytest tests/test_operators_gpu.py tests/test_fp64_anchor_gpu.py tests/test_pwmlp_rows_gpu.py tests/test_pass_calls_gpu.py tests/test_beside_bf16_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

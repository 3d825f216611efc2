#!/bin/bash
# Round 6 session 67: PseudoGrid's channel pairs as scalar FMAs (variant pg_scalar, -DCL3D_PG_PK=0) against the shipped packed pairs:
# the operator step, the config-3 backbone, parity
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s67}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
V=$PWD/scripts/micro/var/libcl3d_pg_scalar.so
line() { grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(sys.argv[1], d.get('ms_per_step'))" "$1"; }
echo "== PseudoGrid operator step, shipped / pg_scalar, alternating" | tee $OUT/summary.txt
for i in 1 2 3; do
  timeout 300 python bench.py --operator pseudo_grid --steps 100 --backbone off --no-cpu-baseline --no-kernel-roofline 2>>$OUT/err.log | line shipped | tee -a $OUT/summary.txt
  CL3D_LIB=$V timeout 300 python bench.py --operator pseudo_grid --steps 100 --backbone off --no-cpu-baseline --no-kernel-roofline 2>>$OUT/err.log | line pg_scalar | tee -a $OUT/summary.txt
done
echo "== config 3 backbone (s3dis_pseudogrid), shipped / pg_scalar" | tee -a $OUT/summary.txt
for i in 1 2; do
  timeout 400 python scripts/bench_backbone.py --config s3dis_pseudogrid --steps 30 2>>$OUT/err.log | line shipped | tee -a $OUT/summary.txt
  CL3D_LIB=$V timeout 400 python scripts/bench_backbone.py --config s3dis_pseudogrid --steps 30 2>>$OUT/err.log | line pg_scalar | tee -a $OUT/summary.txt
done
echo "== parity of the variant" | tee -a $OUT/summary.txt
CL3D_LIB=$V timeout 900 python -m pytest tests/test_operators_gpu.py tests/test_fp64_anchor_gpu.py tests/test_pass_calls_gpu.py tests/test_bottleneck_gpu.py tests/test_beside_bf16_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2 | tee -a $OUT/summary.txt
echo "== the rebuilt shipped library (scalar ball query, scalar TRAIN walk): quick parity + headline" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_bq_paths_gpu.py tests/test_native_gpu.py tests/test_d2_form.py tests/test_operators_gpu.py tests/test_beside_bf16_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2 | tee -a $OUT/summary.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>>$OUT/err.log | grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('driver flags', d['ms_per_step'], d['value'], 'TRAIN', d['roofline']['us'], 'backbone_step', (d.get('backbone_step') or {}).get('ms_per_step'))" | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# Round 6 session 66: the ball query's two distance chains as scalar code (variant bq_scalar, -DCL3D_TL_PK=0) against the shipped packed
# chain: the query alone, the replayed headline step, bit-exactness
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s66}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
V=$PWD/scripts/micro/var/libcl3d_${VARIANT:-bq_scalar}.so
line() { grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print(sys.argv[1], d.get('ms_per_step', d.get('us_median')), d.get('us_min', ''))" "$1"; }
echo "== ball query alone (tile path; mult 1.5 and 4), shipped / variant, alternating" | tee $OUT/summary.txt
for i in 1 2 3; do
  CL3D_BQ_PATH=tile timeout 120 python scripts/bench_bq.py 2>>$OUT/err.log | line shipped | tee -a $OUT/summary.txt
  CL3D_LIB=$V CL3D_BQ_PATH=tile timeout 120 python scripts/bench_bq.py 2>>$OUT/err.log | line variant | tee -a $OUT/summary.txt
done
CL3D_BQ_PATH=tile timeout 120 python scripts/bench_bq.py --mult 4.0 2>>$OUT/err.log | line "shipped mult 4" | tee -a $OUT/summary.txt
CL3D_LIB=$V CL3D_BQ_PATH=tile timeout 120 python scripts/bench_bq.py --mult 4.0 2>>$OUT/err.log | line "variant mult 4" | tee -a $OUT/summary.txt
echo "== headline, shipped / variant, alternating" | tee -a $OUT/summary.txt
for i in 1 2 3; do
  timeout 300 python bench.py --steps 100 --backbone off --no-cpu-baseline --no-kernel-roofline 2>>$OUT/err.log | line shipped | tee -a $OUT/summary.txt
  CL3D_LIB=$V timeout 300 python bench.py --steps 100 --backbone off --no-cpu-baseline --no-kernel-roofline 2>>$OUT/err.log | line variant | tee -a $OUT/summary.txt
done
echo "== bit-exactness of the variant" | tee -a $OUT/summary.txt
CL3D_LIB=$V timeout 900 python -m pytest ${TESTS:-tests/test_bq_paths_gpu.py tests/test_native_gpu.py tests/test_ref_pin_gpu.py tests/test_beside_bf16_gpu.py} -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

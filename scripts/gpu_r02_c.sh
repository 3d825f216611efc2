#!/bin/bash
# round 2, session C: GEMM tile sweep + kernel-level profile of the three point GEMMs; failed tests re-run
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest (previous failures)" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_d2_form.py tests/test_operators_gpu.py -m gpu -q --timeout=900 -p no:cacheprovider -k "bf16 or other_forms or plumbing" -s > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; grep -E "worst per-stage|passed|failed|Error" $OUT/pytest.log | tail -12 | tee -a $OUT/summary.txt
echo "== tile sweep" | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_point_gemm.py --tiles --sweep --reps 30 2>/dev/null | tee $OUT/tiles.jsonl | tee -a $OUT/summary.txt
echo "== min chunks per slice (wgrad), metric shape" | tee -a $OUT/summary.txt
for cps in 4 8 16 32; do
  CL3D_GEMM_MIN_CPS=$cps timeout 300 python scripts/bench_point_gemm.py --reps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('cps $cps', d['mfma_f32']['bwd_weight_us'], d['mfma_bf16']['bwd_weight_us'])" | tee -a $OUT/summary.txt
done
echo "== rocprofv3 of the A/B script" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o ab -- python $R/scripts/bench_point_gemm.py --reps 20 > $R/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
python scripts/kstats.py $OUT/prof/ab_kernel_stats.csv 1 30 | tee -a $OUT/summary.txt
find $OUT -type f -name "*kernel_trace*" -delete 2>/dev/null
find $OUT -type f -size +3M -delete 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# Round 6 session 26: config 2 (bf16) under the profiler -- every gather-pass / rows / BatchNorm launch of a step by (kernel, grid):
# what the deep stages' launches (16 clouds x 16-256 points, 288-1152 channels) cost against the early ones
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s26
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/prof_bb -o bb -- python $R/scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 150 > $R/$OUT/rocprof_bb.log 2>&1)
grep '^{' $OUT/rocprof_bb.log | tail -1 | cut -c1-300 | tee $OUT/summary.txt
T=$(find $OUT/prof_bb -name "bb_kernel_trace.csv" | head -1)
STEPS=$(python - "$T" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t_end = max(int(r["End_Timestamp"]) for r in rows)
print(sum(1 for r in rows if "pwmlp_hit_coeffs_kernel" in r["Kernel_Name"] and int(r["Start_Timestamp"]) >= t_end - 400e6) // 4)
PY
)
echo "steps in the last 400 ms: $STEPS" | tee -a $OUT/summary.txt
for k in pwmlp_support_kernel pwmlp_query_kernel pwmlp_rows pwmlp_hit pwmlp_finalize bn2_ bn_stats bn_apply csr_ gemm_reduce grid_subsample bq_ ball_query maxpool transpose4; do
  echo "== $k by (kernel, grid)" | tee -a $OUT/summary.txt
  python scripts/ktrace_calls.py $T $k --by-grid 400 $STEPS | head -24 | cut -c1-150 | tee -a $OUT/summary.txt
done
rm -rf $OUT/prof_bb
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# Round 6 session 18: kernel table of the config-2 step with measured GEMM plans -- the steady state only (the last 400 ms of
# the trace: the plan measurements of the warm-up are not in it), and the gather passes / pooling / BatchNorm tails by grid.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s18
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/prof_bb -o bb -- python $R/scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 150 > $R/$OUT/rocprof_bb.log 2>&1)
grep '^{' $OUT/rocprof_bb.log | tail -1 | cut -c1-400 | tee $OUT/summary.txt
T=$(find $OUT/prof_bb -name "bb_kernel_trace.csv" | head -1)
STEPS=$(python - "$T" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pwmlp_hit_coeffs_kernel" in r["Kernel_Name"]]
t_end = max(int(r["End_Timestamp"]) for r in csv.DictReader(open(sys.argv[1])))
print(sum(1 for r in rows if int(r["Start_Timestamp"]) >= t_end - 400e6) // 4)
PY
)
echo "steps in the last 400 ms: $STEPS" | tee -a $OUT/summary.txt
python scripts/ktrace_tail.py $T 400 $STEPS 70 | tee $OUT/backbone_steady_state_measured_plans.txt | tee -a $OUT/summary.txt
for k in pwmlp_support_kernel pwmlp_query_kernel maxpool_bwd_kernel bn2_bwd_small_kernel pwmlp_rows_kernel gemm_reduce; do
  echo "== $k by grid" | tee -a $OUT/summary.txt
  python scripts/ktrace_calls.py $T $k --by-grid 400 $STEPS | head -14 | tee -a $OUT/summary.txt
done
rm -rf $OUT/prof_bb
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# Round-3 session G: DP two-graph exchange, weight-gradient GEMM plan sweep, GEMM SQ counters
TAG=${1:-r03g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== dp tests" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests/test_dp_gpu.py tests/test_mfma_gemm_gpu.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -6 $OUT/pytest.log | tee -a $OUT/summary.txt
echo "== weight-gradient plan sweep (metric shape)" | tee -a $OUT/summary.txt
for tile in "2,1" "1,1"; do for sp in 128 256 512; do
  echo -n "tile $tile split $sp: " | tee -a $OUT/summary.txt
  CL3D_GEMM_TILE=$tile CL3D_GEMM_SPLIT=$sp timeout 200 python scripts/bench_point_gemm.py --reps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bwd_weight f32', round(d['mfma_f32']['bwd_weight_us'],1), 'bf16', round(d['mfma_bf16']['bwd_weight_us'],1), '| fwd', round(d['mfma_f32']['fwd_us'],1), 'bwd_data', round(d['mfma_f32']['bwd_data_us'],1))
" | tee -a $OUT/summary.txt
done; done
echo "== SQ counters of the three products (default plan)" | tee -a $OUT/summary.txt
n=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"; do
  n=$((n+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$OUT/gemm_pmc$n -o pmc -- python $R/scripts/bench_point_gemm.py --reps 5 > $R/$OUT/gemm_pmc$n.log 2>&1)
done
python - <<'PY' | tee -a gpurun_out/r03g/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/r03g/gemm_pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "gemm" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-30s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
find $OUT -name "*kernel_trace*" -delete; find $OUT -type f -size +2M -delete
echo "== done" | tee -a $OUT/summary.txt

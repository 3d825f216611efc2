#!/bin/bash
OUT=gpurun_out/r04f
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== two captures: first variant with stage notes" | tee $OUT/summary.txt
timeout 600 python scripts/repro_two_captures.py --config modelnet_small --one baseline 2>&1 | tail -40 | tee -a $OUT/summary.txt
echo "== SQ / TCC counters of the step's three long kernels" | tee -a $OUT/summary.txt
n=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$OUT/pmc$n -o pmc -- python $R/bench.py --steps 6 --warmup 2 --precondition 0 --no-cpu-baseline --no-kernel-roofline --no-step-table > $R/$OUT/pmc$n.log 2>&1)
done
python - <<PY | tee -a $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cl3d::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if not any(x in k for x in ("bq_tile", "pwmlp_support", "pwmlp_query", "csr_", "pwmlp_rows_kernel")): continue
    print(k[:60])
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    %-24s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
find $OUT -name "*kernel_trace*" -delete; find $OUT -type f -size +2M -delete
echo "== done" | tee -a $OUT/summary.txt

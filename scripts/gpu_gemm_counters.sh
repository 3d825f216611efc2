#!/bin/bash
TAG=${1:-r02i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" ; do
  n=$((n+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$OUT/pmc$n -o pmc -- python $R/scripts/bench_point_gemm.py --C 72 --Co 72 --reps 6 > $R/$OUT/pmc$n.log 2>&1)
  python - "$OUT/pmc$n" <<'PY' | tee -a $OUT/summary.txt
import csv, glob, sys, os, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "mfma_gemm" not in name and "Cijk" not in name: continue
        acc[name[:44] + " g" + r["Grid_Size"] + " v" + r["VGPR_Count"] + " lds" + r["LDS_Block_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k.ljust(75), "  ".join(f"{c.replace('SQ_','')}={sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items())))
PY
  find $OUT/pmc$n -type f -size +2M -delete 2>/dev/null
done
echo "== done" | tee -a $OUT/summary.txt

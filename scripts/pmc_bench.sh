#!/bin/bash
# PMC passes over the bench step (separate passes; no trace domains other than kernel-trace).
# Usage: bash scripts/pmc_bench.sh <tag> <operator> "<counters pass 1>" "<counters pass 2>" ...
TAG=$1; OP=$2; shift 2
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
n=0
for pass in "$@"; do
  n=$((n+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc$n -o pmc -- \
     python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-roofline --operator $OP > $O/pmc$n.log 2>&1)
  python - "$O/pmc$n" <<'PY'
import csv, glob, sys, os, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith("cl3d::"): continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k[:60].ljust(60), "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items())))
PY
  find $O/pmc$n -type f -size +2M -delete 2>/dev/null
done

#!/bin/bash
# A/B of tuning variants selected by an environment variable: bench (3 runs each) + the kernel's rocprof average
# usage: gpu_variants.sh TAG ENVVAR "v0 v1 v2 ..." KERNEL_SUBSTRING
TAG=${1:-var}; VAR=$2; VALUES=$3; KERN=$4
OUT=gpurun_out/$TAG
R=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/summary.txt
for v in $VALUES; do
  echo "== $VAR=$v" | tee -a $OUT/summary.txt
  for i in 1 2; do
    env $VAR=$v timeout 600 python bench.py --no-cpu-baseline --no-kernel-roofline --no-step-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  ms_per_step', d['ms_per_step'])" | tee -a $OUT/summary.txt
  done
  (cd /tmp && rm -rf /tmp/vp && env $VAR=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp -o v -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-step-table > /dev/null 2>&1)
  python - <<PY | tee -a $OUT/summary.txt
import csv, glob
for p in glob.glob("/tmp/vp/**/v_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if any(k in r["Name"] for k in "$KERN".split(",")):
            print("  %-60s calls %4s avg %.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
echo "== done" | tee -a $OUT/summary.txt

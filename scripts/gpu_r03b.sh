#!/bin/bash
# SQ counters of the ball-query kernels (tile vs cells) at the metric shape
OUT=gpurun_out/r03b
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  n=$((n+1))
  for p in tile cells; do
    (cd /tmp && CL3D_BQ_PATH=$p timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$OUT/${p}_$n -o pmc -- python $R/scripts/bench_bq.py --reps 10 > $R/$OUT/${p}_$n.log 2>&1)
  done
done
python - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections
for p in ("tile", "cells"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(f"gpurun_out/r03b/{p}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if "bq_" in k or "ball_query" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(p, k)
        for c, v in sorted(cs.items()):
            print("   %-24s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
    for path in glob.glob(f"gpurun_out/r03b/{p}_1/**/*kernel_trace.csv", recursive=True):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if "bq_" in k or "ball_query" in k:
                d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in d.items():
            v.sort(); print("   trace", k, "median us", v[len(v)//2], "n", len(v))
PY
find $OUT -name "*kernel_trace*" -delete; find $OUT -type f -size +2M -delete

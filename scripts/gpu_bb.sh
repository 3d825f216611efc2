#!/bin/bash
# backbone (config 2) A/B: gradient fork on/off, f32 + bf16, and a per-kernel profile of the f32 step
TAG=${1:-bb}
OUT=gpurun_out/$TAG
R=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/summary.txt
for fg in 1 0; do
  echo "== CL3D_FORK_GRADS=$fg" | tee -a $OUT/summary.txt
  CL3D_FORK_GRADS=$fg timeout 300 python scripts/bench_backbone.py --config modelnet_pointwisemlp 2>/dev/null | tail -1 | cut -c1-260 | tee -a $OUT/summary.txt
  CL3D_FORK_GRADS=$fg timeout 300 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | tail -1 | cut -c1-260 | tee -a $OUT/summary.txt
done
timeout 300 python scripts/bench_backbone.py --config s3dis_pseudogrid 2>/dev/null | tail -1 | cut -c1-260 | tee -a $OUT/summary.txt
(cd /tmp && rm -rf /tmp/bbp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bbp -o bb -- python $R/scripts/bench_backbone.py --config modelnet_pointwisemlp --steps 30 > /dev/null 2>&1)
cp $(find /tmp/bbp -name "bb_kernel_stats.csv" | head -1) $OUT/backbone_f32_kernel_stats.csv
python - <<PY | tee -a $OUT/summary.txt
import csv
rows = list(csv.DictReader(open("$OUT/backbone_f32_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:40]:
    print("  %-70s calls %5s avg %6.1f us  %4.1f%%" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
echo "== done" | tee -a $OUT/summary.txt

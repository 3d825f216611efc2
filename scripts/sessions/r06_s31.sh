#!/bin/bash
# Round 6 session 31: session 30's box read the headline at 0.3345 ms (0.2795-0.284 everywhere else this round): the line with its
# step table on this session's box, and DEBUG_HIP_FORCE_GRAPH_QUEUES = 3 against the default, alternating, headline and config 2
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s31}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
line() { grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(sys.argv[1], d.get('ms_per_step'))" "$1"; }
rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | head -30 > $OUT/smi.txt
echo "== headline with its step table" | tee $OUT/summary.txt
timeout 600 python bench.py --steps 100 --no-cpu-baseline --backbone off 2>$OUT/bench.err > $OUT/bench_line.json
python - $OUT/bench_line.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"])
for r in d.get("roofline", {}).get("step", []):
    print("  %-60s %8.1f us" % (str(r.get("kernel"))[:60], r.get("us", 0.0)))
PY
echo "== headline, queues default / 3, alternating" | tee -a $OUT/summary.txt
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 100 --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line "default" | tee -a $OUT/summary.txt
  DEBUG_HIP_FORCE_GRAPH_QUEUES=3 timeout 300 python bench.py --steps 100 --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line "queues=3" | tee -a $OUT/summary.txt
done
echo "== config 2 bf16 / f32, config 3, queues default / 3" | tee -a $OUT/summary.txt
for cfg in "modelnet_pointwisemlp --precision bf16" "modelnet_pointwisemlp" "s3dis_pseudogrid" "s3dis_pospool_deep"; do
  for rep in 1 2; do
    timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 2>/dev/null | line "$cfg default" | tee -a $OUT/summary.txt
    DEBUG_HIP_FORCE_GRAPH_QUEUES=3 timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 2>/dev/null | line "$cfg queues=3" | tee -a $OUT/summary.txt
  done
done
echo "== done" | tee -a $OUT/summary.txt

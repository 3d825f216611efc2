#!/bin/bash
# Round 6 session 30: what session 29's 894 us per step without a running kernel (config 2, bf16: 110 gaps, median 5.9 us) respond to --
# (a) the two gradient products of a contraction forked only above a point count (fused.FORK_MIN_POINTS; stages hold 65 536 / 16 384 /
#     4096 / 1024 / 256 points), (b) the number of queues the HIP runtime lays a captured graph out on (DEBUG_HIP_FORCE_GRAPH_QUEUES)
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s30}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
line() { grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(sys.argv[1], d.get('ms_per_step'))" "$1"; }
echo "== config 2 bf16, forks above a point count" | tee $OUT/summary.txt
for rep in 1 2; do
  for th in 0 2000 5000 20000 70000; do
    timeout 400 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 30 --fork-min-points $th 2>/dev/null | line "fork_min_points=$th" | tee -a $OUT/summary.txt
  done
done
echo "== config 2 bf16, graph queues" | tee -a $OUT/summary.txt
for rep in 1 2; do
  for q in 1 2 3 4; do
    DEBUG_HIP_FORCE_GRAPH_QUEUES=$q timeout 400 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 30 2>/dev/null | line "queues=$q" | tee -a $OUT/summary.txt
  done
done
echo "== headline, graph queues" | tee -a $OUT/summary.txt
for q in default 3 4; do
  if [ $q = default ]; then unset DEBUG_HIP_FORCE_GRAPH_QUEUES; else export DEBUG_HIP_FORCE_GRAPH_QUEUES=$q; fi
  timeout 300 python bench.py --steps 100 --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line "queues=$q" | tee -a $OUT/summary.txt
done
unset DEBUG_HIP_FORCE_GRAPH_QUEUES
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# Round 6 session 10: the class-logit convolution on the engine (replay determinism of the fork-free step), the
# pipelined-pair experiment of bench.py
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s10
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest" | tee $OUT/summary.txt
timeout 2400 python -m pytest tests/test_dp_gpu.py tests/test_operators_gpu.py tests/test_fp64_anchor_gpu.py tests/test_capture_gpu.py -q -m gpu --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -6 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== the fork-free two-graph step, 5 x 1000 replays; forks in both, 2 x 1000" | tee -a $OUT/summary.txt
run() { local label=$1; shift
  timeout 600 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --warmup 1 --head --overlap "$@" 2>/tmp/err.txt | grep '^{' | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$label', {k:(l[k] if len(l[k])<8 else l[k][:6]+['...',len(l[k])]) for k in ('distinct_late','distinct_early')}, 'varying', l['varying_parameters'])" | tee -a $OUT/summary.txt || tail -3 /tmp/err.txt | tee -a $OUT/summary.txt; }
for i in 1 2 3 4 5; do run "none #$i" --overlap-forks none --repeat-check 1000; done
for i in 1 2; do run "both #$i" --overlap-forks both --repeat-check 1000; done
echo "== bench.py --pipelined (two steps as one graph, the next batch's geometry prefetched), alternating with the plain step" | tee -a $OUT/summary.txt
for i in 1 2 3; do
  timeout 600 python bench.py --pipelined --no-cpu-baseline --no-kernel-roofline --backbone off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('headline', d['ms_per_step'], d['value'], '| pipelined pair', d.get('pipelined_pair'))" | cut -c1-400 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

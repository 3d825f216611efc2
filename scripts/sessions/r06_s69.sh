#!/bin/bash
# Round 6 session 69: the closing evidence at HEAD -- 1000 replays of the bf16 config-2 step under two ranks on one device (the set-up that
# showed a different pattern in nearly every replay before the fix), and 1000 launches of the gather pass beside bf16 contractions
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s69}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== two ranks on one device, bf16, 1000 replays of one step (HIP graph, deferred weight gradients)" | tee $OUT/summary.txt
CL3D_BENCH_ONE_DEVICE=1 timeout 900 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --gpus 2 --warmup 1 --gemm-plans model --repeat-check 1000 2>>$OUT/err.log | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('distinct late', d['distinct_late'][:6], 'early', d['distinct_early'][:6], '|', len(d['varying_parameters']), 'parameters vary')" | tee -a $OUT/summary.txt
echo "== one process, bf16 contractions on a second stream, 1000 launches of the gather pass" | tee -a $OUT/summary.txt
for vc in 64 144; do VC=$vc REPS=1000 timeout 300 python scripts/micro/two_stream_pattern.py 2>>$OUT/err.log | grep "wrong sy" | python -c "
import sys, ast
l = sys.stdin.read(); v = ast.literal_eval(l[l.index('['):])
print(l[:l.index(':')], '%d launches, %d with a wrong element' % (len(v), sum(1 for x in v if x)))" | tee -a $OUT/summary.txt; done
echo "== done" | tee -a $OUT/summary.txt

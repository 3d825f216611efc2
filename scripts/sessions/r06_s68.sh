#!/bin/bash
# Round 6 session 68: the KNOWN-BAD layout of the two-graph step (forks in graph B + a warm-up stream of its own: replay-varying early
# gradients in every session of rounds 3-6) showed one pattern in both closing sessions after the packed-operand fix.  Same box, alternating:
# the shipped library against the one from before the fix (scripts/micro/var/libcl3d_head.so, commit a82e610).
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s68}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() { # name
  timeout 400 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --warmup 1 --head --overlap --overlap-forks b --debug-two-graphs other_stream --unsafe --repeat-check ${N:-400} 2>>$OUT/err.log | grep repeat_check | python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['varying_parameters']
print('$1', 'late', d['distinct_late'][:5], 'early', d['distinct_early'][:5], 'varying parameters', len(v), sorted(v)[:4])" | tee -a $OUT/summary.txt
}
echo "== forks in graph B + warm-up stream of its own, ${N:-400} replays, two ranks on one device" | tee $OUT/summary.txt
for i in 1 2; do
  unset CL3D_LIB; run "shipped      #$i"
  export CL3D_LIB=$PWD/scripts/micro/var/libcl3d_head.so; run "before fix   #$i"
done
unset CL3D_LIB
echo "== done" | tee -a $OUT/summary.txt

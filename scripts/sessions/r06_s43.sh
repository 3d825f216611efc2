#!/bin/bash
# Round 6 sessions 43 / 45: the contractions one by one (43), then the PointWiseMLP forward pieces (45), alone and beside a second process
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s43}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python scripts/micro/pwmlp_repeat_under_load.py 2>$OUT/err.log | tee $OUT/summary.txt
tail -5 $OUT/err.log
echo "== done" | tee -a $OUT/summary.txt

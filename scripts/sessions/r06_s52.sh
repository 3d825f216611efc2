#!/bin/bash
# Round 6 session 52: the same interference inside ONE process (two streams)?
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s52}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python scripts/micro/${PROG:-two_stream_interference.py} 2>$OUT/err.log | cut -c1-300 | tee $OUT/summary.txt
tail -3 $OUT/err.log
echo "== done" | tee -a $OUT/summary.txt

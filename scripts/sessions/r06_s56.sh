#!/bin/bash
# Round 6 session 56: which lanes -- the pattern at 144 / 72 channels (18 lanes x 3 groups: 54 of 64 lanes active) and at 64 channels
# (16 lanes x 4 groups: every lane active), one process, bf16 contractions on a second stream
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s56}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for vc in 144 72 64 128; do
  VC=$vc timeout 200 python scripts/micro/two_stream_pattern.py 2>>$OUT/err.log | cut -c1-400 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

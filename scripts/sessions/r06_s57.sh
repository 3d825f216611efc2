#!/bin/bash
# Round 6 sessions 57 / 61: variant libraries of the TRAIN walk (VARIANTS=...) beside bf16 contractions on a second stream: 57 = idling a few cycles
# walk_nop7 / walk_nop1, csrc/fused_pwmlp.hip CL3D_WALK_NOP) remove the wrong elements beside bf16 contractions?
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s57}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 REPS=60
for lib in shipped ${VARIANTS:-walk_nop7 walk_nop1} shipped; do
  if [ $lib = shipped ]; then unset CL3D_LIB; else export CL3D_LIB=$PWD/scripts/micro/var/libcl3d_$lib.so; fi
  for vc in 64 144; do
    echo "-- $lib" | tee -a $OUT/summary.txt
    VC=$vc timeout 200 python scripts/micro/two_stream_pattern.py 2>>$OUT/err.log | grep "wrong sy" | cut -c1-420 | tee -a $OUT/summary.txt
  done
done
echo "== done" | tee -a $OUT/summary.txt

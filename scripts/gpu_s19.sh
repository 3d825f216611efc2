#!/bin/bash
OUT=gpurun_out/r04ac
mkdir -p $OUT
OLD=$PWD/scripts/micro/var/libcl3d_train_scalar.so
echo "== bits: packed TRAIN walk against the scalar walk" | tee $OUT/summary.txt
python scripts/micro/ab_bits.py --out /tmp/new.pt 2>&1 | tail -5 | tee -a $OUT/summary.txt
CL3D_LIB=$OLD python scripts/micro/ab_bits.py --out /tmp/old.pt 2>&1 | tail -5 | tee -a $OUT/summary.txt
python scripts/micro/ab_bits.py --compare /tmp/new.pt /tmp/old.pt 2>&1 | tail -5 | tee -a $OUT/summary.txt

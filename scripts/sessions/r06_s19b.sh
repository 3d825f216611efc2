#!/bin/bash
# Round 6 session 19b: lanes per row for the deeper stages' widths (C = 288 / 576 / 1152 on 256 / 64 / 16 points per cloud)
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s19
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
export CL3D_LIB=$R/scripts/micro/var/libcl3d_pw_lanes_env.so
show() { grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = {r['entry']: r['us'] for r in d['roofline']['step']['kernels']}
print(sys.argv[1], 'ms', d['ms_per_step'], ' '.join('%s=%.1f' % (n.replace('cl3d_pwmlp_', ''), k[n]) for n in k if 'stats' in n or 'support' in n or 'rows' in n or 'hit' in n))" "$1"; }
echo "== C = 288, 16 x 256 points (default 12 lanes x 5, 6 chunks)" | tee $OUT/summary_b.txt
for L in "" 18 24 32 36; do
  CL3D_LANES=$L timeout 300 python bench.py --channels 288 --points 256 --no-cpu-baseline --backbone off 2>$OUT/err.log | show "lanes=${L:-auto}" | tee -a $OUT/summary_b.txt
done
echo "== C = 576, 16 x 64 points (default 16 lanes x 4, 9 chunks)" | tee -a $OUT/summary_b.txt
for L in "" 24 32 48 64; do
  CL3D_LANES=$L timeout 300 python bench.py --channels 576 --points 64 --nsample 16 --no-cpu-baseline --backbone off 2>$OUT/err.log | show "lanes=${L:-auto}" | tee -a $OUT/summary_b.txt
done
echo "== C = 36, 16 x 4096 points (default ?)" | tee -a $OUT/summary_b.txt
for L in "" 9 12 16; do
  CL3D_LANES=$L timeout 300 python bench.py --channels 36 --no-cpu-baseline --backbone off 2>$OUT/err.log | show "lanes=${L:-auto}" | tee -a $OUT/summary_b.txt
done
tail -3 $OUT/err.log | cut -c1-300 | tee -a $OUT/summary_b.txt
echo "== done" | tee -a $OUT/summary_b.txt

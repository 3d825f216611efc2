#!/bin/bash
# Round 6 session 19: how the PointWiseMLP gather passes cut a 72- / 144-channel row into lanes (CL3D_LANES, variant build):
# fewer, wider rows per wave-load against more queries per wave.  Operator step at the config-2 stage shapes.
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
echo "== C = 72, 16 x 4096 points (default: 9 lanes x 7 queries, 2 chunks)" | tee $OUT/summary.txt
for L in "" 18 12 9 6; do
  CL3D_LANES=$L timeout 300 python bench.py --channels 72 --no-cpu-baseline --backbone off 2>$OUT/err.log | show "lanes=${L:-auto}" | tee -a $OUT/summary.txt
done
echo "== C = 144, 16 x 1024 points (default: 12 lanes x 5 queries, 3 chunks)" | tee -a $OUT/summary.txt
for L in "" 36 18 12 9; do
  CL3D_LANES=$L timeout 300 python bench.py --channels 144 --points 1024 --no-cpu-baseline --backbone off 2>$OUT/err.log | show "lanes=${L:-auto}" | tee -a $OUT/summary.txt
done
echo "== C = 64 (the headline; default 16 x 4)" | tee -a $OUT/summary.txt
for L in "" 16 8; do
  CL3D_LANES=$L timeout 300 python bench.py --no-cpu-baseline --backbone off 2>$OUT/err.log | show "lanes=${L:-auto}" | tee -a $OUT/summary.txt
done
tail -3 $OUT/err.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

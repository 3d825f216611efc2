#!/bin/bash
# Round 6 session 28: the support-major pass walking its channel chunks INSIDE the tile (docs/experiments/r06_support_chunks_inside_the_tile.patch; on top of session 27)
# (gx x chunks <= 1024) against the tree at HEAD (two generations of 1-2 tiles), alternating runs
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s28}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
echo "== pytest" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_pwmlp_support_gpu.py tests/test_operators_gpu.py tests/test_pwmlp_rows_gpu.py tests/test_pass_calls_gpu.py tests/test_config2_fullsize_gpu.py tests/test_bottleneck_gpu.py tests/test_abi_host_gpu.py -x -q -m gpu --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest.log | cut -c1-400 | tee -a $OUT/summary.txt
line() { grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(sys.argv[1], d.get('ms_per_step'))" "$1"; }
for cfg in "modelnet_pointwisemlp --precision bf16" "modelnet_pointwisemlp"; do
  echo "== backbone $cfg, new / head" | tee -a $OUT/summary.txt
  for i in 1 2 3 4; do
    (cd $R && timeout 400 python scripts/bench_backbone.py --config $cfg 2>/dev/null | line new | tee -a $OUT/summary.txt)
    (cd $R/scripts/micro/var/head_tree && timeout 400 python scripts/bench_backbone.py --config $cfg 2>/dev/null | line head | tee -a $R/$OUT/summary.txt)
  done
done
echo "== headline (one chunk: unchanged grids), new / head" | tee -a $OUT/summary.txt
(cd $R && timeout 300 python bench.py --steps 100 --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line new | tee -a $OUT/summary.txt)
(cd $R/scripts/micro/var/head_tree && timeout 300 python bench.py --steps 100 --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line head | tee -a $R/$OUT/summary.txt)
echo "== the 72-, 144- and 288-channel operator steps, new / head" | tee -a $OUT/summary.txt
for ch in 72 144 288; do
  (cd $R && timeout 300 python bench.py --steps 100 --channels $ch --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line "new $ch" | tee -a $OUT/summary.txt)
  (cd $R/scripts/micro/var/head_tree && timeout 300 python bench.py --steps 100 --channels $ch --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line "head $ch" | tee -a $R/$OUT/summary.txt)
done
echo "== done" | tee -a $OUT/summary.txt

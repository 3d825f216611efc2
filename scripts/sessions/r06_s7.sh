#!/bin/bash
# Round 6 session 7: PseudoGrid backward with its kernel-weight pass forked (A/B against the variant without the fork),
# CSR loops unrolled (configs 3 / 5, metric-shape CSR), tests of the touched files
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s7
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
echo "== pytest" | tee $OUT/summary.txt
timeout 2400 python -m pytest tests/test_operators_gpu.py tests/test_scene_size_gpu.py tests/test_pass_calls_gpu.py tests/test_bottleneck_gpu.py tests/test_fp64_anchor_gpu.py tests/test_capture_gpu.py -q -m gpu --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -6 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== PseudoGrid step, kernel-weight pass forked / not (variant library), alternating" | tee -a $OUT/summary.txt
for i in 1 2 3; do
  timeout 600 python bench.py --operator pseudo_grid --no-cpu-baseline --no-kernel-roofline --backbone off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('forked    ', d['ms_per_step'], d['value'])" | tee -a $OUT/summary.txt
  CL3D_LIB=$R/scripts/micro/var/libcl3d_pg_nofork.so timeout 600 python bench.py --operator pseudo_grid --no-cpu-baseline --no-kernel-roofline --backbone off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('one stream', d['ms_per_step'], d['value'])" | tee -a $OUT/summary.txt
done
for i in 1 2; do
  timeout 600 python scripts/bench_backbone.py --config s3dis_pseudogrid 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config 3 forked    ', d['ms_per_step'])" | tee -a $OUT/summary.txt
  CL3D_LIB=$R/scripts/micro/var/libcl3d_pg_nofork.so timeout 600 python scripts/bench_backbone.py --config s3dis_pseudogrid 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config 3 one stream', d['ms_per_step'])" | tee -a $OUT/summary.txt
done
timeout 600 python bench.py --operator pseudo_grid --no-graph --no-cpu-baseline --no-kernel-roofline --backbone off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('eager forked', d['ms_per_step'], d['value'])" | tee -a $OUT/summary.txt
echo "== config 5 / headline" | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_backbone.py --config s3dis_pospool_deep 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config 5', d['ms_per_step'])" | tee -a $OUT/summary.txt
timeout 600 python bench.py --no-cpu-baseline --no-kernel-roofline --backbone off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('headline', d['ms_per_step'], d['value'])" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c5 -o bb -- python $R/scripts/bench_backbone.py --config s3dis_pospool_deep --steps 20 > $R/$OUT/rocprof_c5.log 2>&1)
python scripts/kstats.py $(find $OUT/prof_c5 -name "bb_kernel_stats.csv" | head -1) 27 60 | grep -i "csr\|grid_s\|sort\|total" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_pg -o bench -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-roofline --backbone off --operator pseudo_grid > $R/$OUT/rocprof_pg.log 2>&1)
python scripts/kstats.py $(find $OUT/prof_pg -name "bench_kernel_stats.csv" | head -1) 345 12 | tee -a $OUT/summary.txt
find $OUT -name "*kernel_trace*" -delete 2>/dev/null; find $OUT -type f -size +3M -delete 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt

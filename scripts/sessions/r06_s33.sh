#!/bin/bash
# Round 6 session 33: the few-slice weight-gradient reduce (gemm_reduce_w4_kernel: one thread per four input channels, every thread
# storing) and four slices in flight in the other slice sums -- parity suites, then config 2 against the library at HEAD
# (scripts/micro/var/libcl3d_head.so), alternating, with the runtime's default graph layout and with DEBUG_HIP_FORCE_GRAPH_QUEUES=3
# (sessions 30 / 31: config 2 bf16 6.01 -> 5.78 ms on those boxes); the headline under both layouts on this box
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s33}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
line() { grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(sys.argv[1], d.get('ms_per_step'))" "$1"; }
echo "== pytest" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_mfma_gemm_gpu.py tests/test_operators_gpu.py tests/test_bottleneck_gpu.py tests/test_config2_fullsize_gpu.py tests/test_pass_calls_gpu.py tests/test_pwmlp_rows_gpu.py -x -q -m gpu --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest.log | cut -c1-400 | tee -a $OUT/summary.txt
for q in default 3; do
  if [ $q = default ]; then unset DEBUG_HIP_FORCE_GRAPH_QUEUES; else export DEBUG_HIP_FORCE_GRAPH_QUEUES=$q; fi
  for cfg in "modelnet_pointwisemlp --precision bf16" "modelnet_pointwisemlp"; do
    echo "== backbone $cfg, queues=$q, new / head" | tee -a $OUT/summary.txt
    for i in 1 2 3; do
      timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 2>/dev/null | line new | tee -a $OUT/summary.txt
      CL3D_LIB=$R/scripts/micro/var/libcl3d_head.so timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 2>/dev/null | line head | tee -a $OUT/summary.txt
    done
  done
done
unset DEBUG_HIP_FORCE_GRAPH_QUEUES
echo "== headline, queues default / 3 / 4, alternating" | tee -a $OUT/summary.txt
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 200 --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line "default" | tee -a $OUT/summary.txt
  DEBUG_HIP_FORCE_GRAPH_QUEUES=3 timeout 300 python bench.py --steps 200 --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line "queues=3" | tee -a $OUT/summary.txt
  DEBUG_HIP_FORCE_GRAPH_QUEUES=4 timeout 300 python bench.py --steps 200 --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line "queues=4" | tee -a $OUT/summary.txt
done
echo "== the other operators and configs 3 / 4 / 5, queues default / 3" | tee -a $OUT/summary.txt
for op in pospool adaptive_weight pseudo_grid; do
  timeout 300 python bench.py --operator $op --steps 200 --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line "$op default" | tee -a $OUT/summary.txt
  DEBUG_HIP_FORCE_GRAPH_QUEUES=3 timeout 300 python bench.py --operator $op --steps 200 --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line "$op queues=3" | tee -a $OUT/summary.txt
done
for cfg in s3dis_pseudogrid partnet_adaptive s3dis_pospool_deep; do
  timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 2>/dev/null | line "$cfg default" | tee -a $OUT/summary.txt
  DEBUG_HIP_FORCE_GRAPH_QUEUES=3 timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 2>/dev/null | line "$cfg queues=3" | tee -a $OUT/summary.txt
done
echo "== config 2 bf16 slice-sum rows, new" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bb -- python $R/scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 40 > /dev/null 2>&1)
S=$(find $OUT/prof -name "bb_kernel_stats.csv" | head -1)
[ -n "$S" ] && grep -E "gemm_reduce" $S | cut -c1-160 | tee -a $OUT/summary.txt
rm -rf $OUT/prof
echo "== done" | tee -a $OUT/summary.txt

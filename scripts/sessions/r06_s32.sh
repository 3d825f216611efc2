#!/bin/bash
# Round 6 session 32: (a) sessions 29-31 read the headline at 0.33 ms with only the ball query (x1.54) and the TRAIN pass (x1.35) slower than
# in the closing session and every memory-bound kernel unchanged: the shader clock under load, sampled while the bench runs;
# (b) bn2_{fwd,bwd}_small_kernel with one 1024-thread workgroup per channel for 8192 < B N <= 16384 (config 2's 288-channel layers:
# 33 us per backward launch on 256 threads): tests, then config 2 against the library at HEAD (scripts/micro/var/libcl3d_head.so), alternating
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s32}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
line() { grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(sys.argv[1], d.get('ms_per_step'))" "$1"; }
echo "== clocks while the headline runs" | tee $OUT/summary.txt
(for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo; sleep 0.5; done) > $OUT/clocks.txt &
SMI=$!
timeout 300 python bench.py --steps 20000 --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line "headline (20000 steps)" | tee -a $OUT/summary.txt
wait $SMI
sort $OUT/clocks.txt | uniq -c | sort -rn | head -8 | tee -a $OUT/summary.txt
rocm-smi --showmaxpower --showperflevel --showvoltage 2>/dev/null | grep -E "GPU\[" | tee -a $OUT/summary.txt
echo "== pytest" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests/test_operators_gpu.py tests/test_bottleneck_gpu.py tests/test_config2_fullsize_gpu.py -x -q -m gpu --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest.log | cut -c1-400 | tee -a $OUT/summary.txt
for cfg in "modelnet_pointwisemlp --precision bf16" "modelnet_pointwisemlp"; do
  echo "== backbone $cfg, new / head" | tee -a $OUT/summary.txt
  for i in 1 2 3; do
    timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 2>/dev/null | line new | tee -a $OUT/summary.txt
    CL3D_LIB=$R/scripts/micro/var/libcl3d_head.so timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 2>/dev/null | line head | tee -a $OUT/summary.txt
  done
done
echo "== config 2 bf16 kernel table, new" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bb -- python $R/scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 40 > /dev/null 2>&1)
S=$(find $OUT/prof -name "bb_kernel_stats.csv" | head -1)
[ -n "$S" ] && grep -E "bn2_|bn_stats|bn_apply" $S | cut -c1-160 | tee -a $OUT/summary.txt
rm -rf $OUT/prof
echo "== done" | tee -a $OUT/summary.txt

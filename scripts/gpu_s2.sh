#!/bin/bash
# round 5, session 2: layout A of the captured step (CSR behind the stats pass, ordered behind the product)
OUT=gpurun_out/r05c
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bm() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms_per_step', d['ms_per_step'], d['config']['launch'])"; }
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>$OUT/err$i.txt | bm "layout A" | tee -a $OUT/summary.txt
  tail -3 $OUT/err$i.txt | cut -c1-300 | tee -a $OUT/summary.txt
done
CL3D_CSR_FIRST=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "r4 order + one grad kernel" | tee -a $OUT/summary.txt
DEBUG_HIP_FORCE_GRAPH_QUEUES=3 timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "layout A, 3 graph queues" | tee -a $OUT/summary.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "driver flags" | tee -a $OUT/summary.txt
timeout 300 python bench.py --no-graph --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "eager" | tee -a $OUT/summary.txt
for op in pospool adaptive_weight pseudo_grid; do
  timeout 300 python bench.py --operator $op --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "$op" | tee -a $OUT/summary.txt
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-kernel-roofline --precondition 0 --steps 50 > $R/$OUT/rocprof.log 2>&1)
python scripts/step_timeline.py "$OUT/prof/**/bench_kernel_trace.csv" | tee $OUT/step_timeline.txt | tee -a $OUT/summary.txt
timeout 600 python -m pytest tests/test_operators_gpu.py tests/test_pass_calls_gpu.py tests/test_dp_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | cut -c1-250 | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp 2>/dev/null | tail -1 | cut -c1-300 | tee -a $OUT/summary.txt

#!/bin/bash
OUT=gpurun_out/r05e
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bm() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms_per_step', d['ms_per_step'], d['config']['launch'])"; }
timeout 600 python -m pytest tests/test_mfma_gemm_gpu.py -k "both_gradients" -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | cut -c1-250 | tee -a $OUT/summary.txt
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "layout A + dw reduce" | tee -a $OUT/summary.txt
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "driver flags" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-kernel-roofline --precondition 0 --steps 50 > $R/$OUT/rocprof.log 2>&1)
python scripts/step_timeline.py "$OUT/prof/**/bench_kernel_trace.csv" | tee $OUT/step_timeline.txt | tail -8 | tee -a $OUT/summary.txt

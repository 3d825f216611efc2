#!/bin/bash
# round 5, session 1: the one-kernel gradient pair + the capture order of the step (CSR build behind the first consumer)
OUT=gpurun_out/r05a
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mfma_gemm_gpu.py tests/test_pass_calls_gpu.py tests/test_operators_gpu.py tests/test_bottleneck_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | cut -c1-250 | tee $OUT/summary.txt
bm() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms_per_step', d['ms_per_step'])"; }
for i in 1 2; do
  CL3D_CSR_FIRST=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "csr-first(r4 order)" | tee -a $OUT/summary.txt
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "csr-behind-stats" | tee -a $OUT/summary.txt
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "driver flags" | tee -a $OUT/summary.txt
timeout 300 python bench.py --no-graph --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "eager" | tee -a $OUT/summary.txt
for v in "" "CL3D_CSR_FIRST=1"; do
  (cd /tmp && env $v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof$v -o bench -- python $R/bench.py --no-cpu-baseline --no-kernel-roofline --precondition 0 --steps 50 > $R/$OUT/rocprof$v.log 2>&1)
  echo "== timeline [$v]" | tee -a $OUT/summary.txt
  python scripts/step_timeline.py "$OUT/prof$v/**/bench_kernel_trace.csv" | tee $OUT/step_timeline$v.txt | tee -a $OUT/summary.txt
done
python scripts/kstats.py $OUT/prof/bench_kernel_stats.csv 60 30 2>/dev/null | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_point_gemm.py 2>/dev/null | cut -c1-900 | tee -a $OUT/summary.txt

#!/bin/bash
# One GPU-box session: parity tests, reference pin, native golden vectors, bench, rocprof summary.
# Usage (from the repo root on the GPU box): bash scripts/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== smoke" | tee $OUT/summary.txt
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
echo "== pytest gpu (engine vs oracle)" | tee -a $OUT/summary.txt
timeout 1200 python -m pytest tests/test_native_gpu.py tests/test_operators_gpu.py -m gpu -q --timeout=300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
echo "== reference pin (oracle/_ref)" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_ref_pin_gpu.py -m gpu -q --timeout=300 -p no:cacheprovider > $OUT/pytest_refpin.log 2>&1
echo "refpin rc=$?" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest_refpin.log | tee -a $OUT/summary.txt
echo "== native golden from the reference kernels" | tee -a $OUT/summary.txt
timeout 600 python tests/golden/make_native_golden.py $OUT/native_golden > $OUT/native_golden.log 2>&1; echo "golden rc=$?" | tee -a $OUT/summary.txt
echo "== bench" | tee -a $OUT/summary.txt
timeout 900 python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench.json | tee -a $OUT/summary.txt
echo "== rocprofv3 kernel trace of the same bench command" | tee -a $OUT/summary.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof -name '*stats*' | head -5 | tee -a $OUT/summary.txt
python scripts/kstats.py $OUT/prof/bench_kernel_stats.csv 60 30 | tee -a $OUT/summary.txt
# keep the merged output small: drop raw traces, keep stats
find $OUT/prof -type f ! -name "*stats*" -size +1M -delete 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt

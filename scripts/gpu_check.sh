#!/bin/bash
# One GPU-box session: parity tests, reference pin, native golden vectors, bench (+ rocprof summary of the same
# command), PMC traffic of the ball_query+group kernels, the other operators and the backbone configs.
# Usage (from the repo root on the GPU box): bash scripts/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== smoke" | tee $OUT/summary.txt
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
echo "== pytest -m gpu (engine vs oracle, golden fixtures, reference pin, full-size properties)" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
echo "== native golden from the reference kernels" | tee -a $OUT/summary.txt
timeout 600 python tests/golden/make_native_golden.py $OUT/native_golden > $OUT/native_golden.log 2>&1; echo "golden rc=$?" | tee -a $OUT/summary.txt
echo "== bench (the driver's command, default flags)" | tee -a $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench.json | tee -a $OUT/summary.txt
echo "== bench, eager launches" | tee -a $OUT/summary.txt
timeout 900 python bench.py --no-graph --no-cpu-baseline --no-kernel-roofline 2>/dev/null | tee $OUT/bench_eager.json | cut -c1-260 | tee -a $OUT/summary.txt
echo "== rocprofv3 kernel trace of the same bench command" | tee -a $OUT/summary.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
python scripts/kstats.py $OUT/prof/bench_kernel_stats.csv 36 40 | tee -a $OUT/summary.txt
echo "== PMC traffic of the ball_query+group kernels (separate passes)" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_fetch -o pmc -- python $R/scripts/pmc_kernels.py > $R/$OUT/pmc_fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_write -o pmc -- python $R/scripts/pmc_kernels.py > $R/$OUT/pmc_write.log 2>&1)
python scripts/pmc_kernels.py --parse $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json 2>> $OUT/summary.txt
head -c 1500 $OUT/pmc_traffic.json | tee -a $OUT/summary.txt
echo "== other operators (bench.py --operator)" | tee -a $OUT/summary.txt
for op in pospool adaptive_weight pseudo_grid; do
  timeout 600 python bench.py --operator $op --no-cpu-baseline --no-kernel-roofline 2>/dev/null | tee $OUT/bench_$op.json | cut -c1-330 | tee -a $OUT/summary.txt
done
echo "== backbone steps (scripts/bench_backbone.py)" | tee -a $OUT/summary.txt
for c in modelnet_small modelnet_pointwisemlp s3dis_pseudogrid partnet_adaptive s3dis_pospool_deep; do
  timeout 600 python scripts/bench_backbone.py --config $c 2>/dev/null | tail -1 | tee -a $OUT/summary.txt
done
echo "== dataset-side grid subsampling (SURVEY 8(f) rank 2): engine vs the reference's C++ on the host" | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_dataset_grid.py 2>/dev/null | tee $OUT/bench_dataset_grid.json | tee -a $OUT/summary.txt
echo "== vote bookkeeping and sphere crops on the device (SURVEY 8(f) ranks 3 and 2)" | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_voting.py 2>/dev/null | tee $OUT/bench_voting.json | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_sphere_crop.py 2>/dev/null | tee $OUT/bench_sphere_crop.json | tee -a $OUT/summary.txt
# keep the merged output small: drop raw traces, keep stats and counter tables
find $OUT -type f -name "*kernel_trace*" -delete 2>/dev/null
find $OUT -type f -size +3M -delete 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# round 2, session A: the new MFMA contraction (tests, A/B vs the library, bench line with the step table) and a fresh
# steady-state kernel table of the config-2 backbone step.
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest: MFMA GEMM + operators" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_mfma_gemm_gpu.py tests/test_operators_gpu.py -m gpu -q --timeout=600 -p no:cacheprovider -x > $OUT/pytest_a.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -15 $OUT/pytest_a.log | tee -a $OUT/summary.txt
echo "== A/B point GEMM" | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_point_gemm.py --sweep 2>&1 | tee $OUT/point_gemm.jsonl | cut -c1-1200 | tee -a $OUT/summary.txt
echo "== bench (default flags)" | tee -a $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench.json | tee -a $OUT/summary.txt; tail -5 $OUT/bench.err | tee -a $OUT/summary.txt
echo "== bench --precision bf16" | tee -a $OUT/summary.txt
timeout 600 python bench.py --precision bf16 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_bf16.json | cut -c1-400 | tee -a $OUT/summary.txt
echo "== rocprofv3 of the bench command" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-kernel-roofline > $R/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
python scripts/kstats.py $OUT/prof/bench_kernel_stats.csv 113 30 | tee -a $OUT/summary.txt
echo "== backbone steps" | tee -a $OUT/summary.txt
for c in modelnet_pointwisemlp s3dis_pseudogrid; do
  timeout 600 python scripts/bench_backbone.py --config $c 2>/dev/null | tail -1 | tee -a $OUT/summary.txt
done
echo "== rocprofv3 of the config-2 backbone step (40 replays)" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_bb -o bb -- python $R/scripts/bench_backbone.py --config modelnet_pointwisemlp --steps 40 > $R/$OUT/rocprof_bb.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
python scripts/kstats.py $OUT/prof_bb/bb_kernel_stats.csv 47 60 | tee -a $OUT/summary.txt
find $OUT -type f -name "*kernel_trace*" -delete 2>/dev/null
find $OUT -type f -size +3M -delete 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt

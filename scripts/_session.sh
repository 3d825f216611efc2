set -x
OUT=gpurun_out/r05s1; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_pwmlp_rows_gpu.py -x -q -p no:cacheprovider > $OUT/rows.log 2>&1; echo "rows rc=$?" ; tail -5 $OUT/rows.log
timeout 600 python bench.py --no-cpu-baseline --backbone off > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --no-cpu-baseline --backbone off --precondition 0 > $R/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?"
python scripts/kstats.py $(find $OUT/prof -name "bench_kernel_stats.csv" | head -1) 100 30 > $OUT/kstats.txt; python scripts/step_timeline.py "$OUT/prof/**/bench_kernel_trace.csv" | tee $OUT/step_timeline.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log | cut -c1-300
rm -rf $OUT/prof

OUT=gpurun_out/r05zz; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof2 -o bench -- python $R/bench.py --no-cpu-baseline --backbone off --precondition 0 > $R/$OUT/rocprof2.log 2>&1); echo "rocprof rc=$?"
python scripts/step_timeline.py "$OUT/prof2/**/bench_kernel_trace.csv" | tee $OUT/step_timeline.txt
rm -rf $OUT/prof2

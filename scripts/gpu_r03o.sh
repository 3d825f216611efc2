#!/bin/bash
# kernel tables of the bench commands without the untimed preconditioning steps in the trace; anchor test re-run
TAG=${1:-r03o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_fp64_anchor_gpu.py -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/summary.txt; tail -2 $OUT/pytest.log | tee -a $OUT/summary.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --no-cpu-baseline --precondition 0 > $R/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
python scripts/kstats.py $(find $OUT/prof -name "bench_kernel_stats.csv" | head -1) 100 40 | tee -a $OUT/summary.txt
(cd /tmp && rm -rf /tmp/pgp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pgp -o pg -- python $R/bench.py --operator pseudo_grid --steps 40 --warmup 5 --precondition 0 --no-cpu-baseline --no-kernel-roofline --no-step-table > /dev/null 2>&1)
cp $(find /tmp/pgp -name "pg_kernel_stats.csv" | head -1) $OUT/bench_pseudo_grid_kernel_stats.csv
python scripts/kstats.py $OUT/bench_pseudo_grid_kernel_stats.csv 45 12 | tee -a $OUT/summary.txt

#!/bin/bash
# parity subset for the fused PointWiseMLP path + bench x3 + per-kernel rocprof averages (bounded timeouts)
TAG=${1:-sq}
OUT=gpurun_out/$TAG
R=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_operators_gpu.py tests/test_native_gpu.py tests/test_abi_host_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout=300 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/summary.txt; grep -E "passed|failed|^FAILED|Error:|assert" $OUT/pytest.log | tail -8 | tee -a $OUT/summary.txt
for i in 1 2 3; do
timeout 120 python bench.py --no-cpu-baseline --no-kernel-roofline --no-step-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  ms_per_step', d['ms_per_step'])" | tee -a $OUT/summary.txt
done
(cd /tmp && rm -rf /tmp/vp && timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp -o v -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-step-table > /dev/null 2>&1)
python - <<PY | tee -a $OUT/summary.txt
import csv, glob
for p in glob.glob("/tmp/vp/**/v_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(p)))[:22]:
        print("  %-62s calls %4s avg %6.1f us" % (r["Name"].split("(")[0][-62:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
echo "== done" | tee -a $OUT/summary.txt

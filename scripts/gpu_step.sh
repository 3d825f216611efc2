#!/bin/bash
# headline step: parity subset, bench x3, kernel-trace timeline of one graph replay (scripts/step_timeline.py)
TAG=${1:-step}
OUT=gpurun_out/$TAG
R=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest (operators, native, abi host, fullsize)" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests/test_operators_gpu.py tests/test_native_gpu.py tests/test_abi_host_gpu.py tests/test_fullsize_gpu.py tests/test_ref_pin_gpu.py -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; grep -E "passed|failed|^FAILED|Error:" $OUT/pytest.log | tail -8 | tee -a $OUT/summary.txt
for i in 1 2 3; do
timeout 600 python bench.py --no-cpu-baseline --no-kernel-roofline --no-step-table 2>/dev/null | cut -c1-200 | tee -a $OUT/summary.txt
done
timeout 600 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null > $OUT/bench_step.json
python - <<PY | tee -a $OUT/summary.txt
import json
d = json.load(open("$OUT/bench_step.json"))
s = d["roofline"]["step"]
print("eager sum %.1f us, graph step %.1f us" % (s["sum_us"], s["graph_step_us"]))
for k in s["kernels"]:
    print("  %-40s %7.2f us" % (k["entry"], k["us"]))
PY
echo "== timeline of one replay" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-roofline --no-step-table > $R/$OUT/tl.log 2>&1)
python scripts/step_timeline.py "/tmp/tl/**/tl_kernel_trace.csv" | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# PseudoGrid: parity tests, operator bench + per-kernel averages, config 3 backbone
TAG=${1:-pg}
OUT=gpurun_out/$TAG
R=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_operators_gpu.py tests/test_fullsize_gpu.py tests/test_scene_size_gpu.py tests/test_abi_host_gpu.py -m gpu -q --timeout=300 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/summary.txt; grep -E "passed|failed|^FAILED|Error:|assert " $OUT/pytest.log | tail -12 | tee -a $OUT/summary.txt
for i in 1 2; do
timeout 120 python bench.py --operator pseudo_grid --no-cpu-baseline --no-kernel-roofline --no-step-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  pseudo_grid ms_per_step', d['ms_per_step'], 'Mpts/s', round(d['value']/1e6,1))" | tee -a $OUT/summary.txt
done
(cd /tmp && rm -rf /tmp/vp && timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp -o v -- python $R/bench.py --operator pseudo_grid --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-step-table > /dev/null 2>&1)
python - <<PY | tee -a $OUT/summary.txt
import csv, glob
for p in glob.glob("/tmp/vp/**/v_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(p)))[:12]:
        print("  %-62s calls %4s avg %6.1f us" % (r["Name"].split("(")[0][-62:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
timeout 300 python scripts/bench_backbone.py --config s3dis_pseudogrid 2>/dev/null | tail -1 | cut -c1-260 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

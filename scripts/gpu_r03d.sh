#!/bin/bash
# Round-3 session D: the new parity tests, timeline of the replayed step, PMC counters of the step kernels
TAG=${1:-r03d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== new parity tests" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests/test_fp64_anchor_gpu.py tests/test_config2_fullsize_gpu.py tests/test_scene_size_gpu.py -m gpu -q -s --timeout=900 -p no:cacheprovider > $OUT/pytest_new.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; grep -v "^$" $OUT/pytest_new.log | tail -60 | tee -a $OUT/summary.txt
echo "== timeline" | tee -a $OUT/summary.txt
rm -rf /tmp/tl
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-roofline > /dev/null 2>&1)
python scripts/step_timeline.py "/tmp/tl/**/tl_kernel_trace.csv" | tee $OUT/step_timeline.txt | tee -a $OUT/summary.txt
echo "== PMC counters of the timed step (separate passes)" | tee -a $OUT/summary.txt
n=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"; do
  n=$((n+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$OUT/step_pmc$n -o pmc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-roofline > $R/$OUT/step_pmc$n.log 2>&1)
done
python scripts/step_counters.py $OUT/step_pmc1 $OUT/step_pmc2 $OUT/step_pmc3 $OUT/step_pmc4 > $OUT/step_counters.json 2>> $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
d = json.load(open("$OUT/step_counters.json"))["kernels"]
for k, v in d.items():
    print("%-60s hbm %7.1f MB  l2req %6.2f M  hit %.2f  valu %6.2f M  wait %.2f" % (k[:60], v.get("hbm_bytes", 0) / 1e6, v.get("l2_bytes", 0) / 128e6, v.get("l2_hit_rate", 0), v.get("SQ_INSTS_VALU", 0) / 1e6, v.get("wait_frac", 0)))
PY
find $OUT -type f -name "*kernel_trace*" -delete 2>/dev/null
find $OUT -type f -size +3M -delete 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt

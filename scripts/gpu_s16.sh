#!/bin/bash
OUT=gpurun_out/r04y
mkdir -p $OUT
export TMPDIR=/tmp
echo "== whole GPU suite" | tee $OUT/summary.txt
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/suite.log 2>&1
echo "suite rc=$?" | tee -a $OUT/summary.txt
tail -4 $OUT/suite.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== eager bench (pass calls), twice; graph" | tee -a $OUT/summary.txt
for flags in "--no-graph" "--no-graph" ""; do
  timeout 300 python bench.py $flags --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flags [$flags]', 'ms_per_step', d['ms_per_step'], d['config']['launch'])" | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

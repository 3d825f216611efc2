#!/bin/bash
OUT=gpurun_out/r04x
mkdir -p $OUT
export TMPDIR=/tmp
echo "== classifier unit through the engine: repeated replays; ordering probe on the forks of graph B" | tee $OUT/summary.txt
i=0
for cfg in "--overlap --overlap-forks none" "--overlap --overlap-forks a" "" "--overlap --overlap-forks b" "--overlap --overlap-forks b --fork-mode probe" "--overlap --overlap-forks both --fork-mode probe" "--overlap --overlap-forks a --fork-mode probe"; do
  i=$((i+1))
  timeout 300 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --warmup 1 --head $cfg --repeat-check 200 > $OUT/run_$i.log 2>&1
  echo "[$cfg] rc=$? $(grep repeat_check $OUT/run_$i.log | cut -c1-700)" | tee -a $OUT/summary.txt
  grep -i "error\|Traceback" $OUT/run_$i.log | head -3 | tee -a $OUT/summary.txt
done
echo "== dp test + operators" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_operators_gpu.py tests/test_capture_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

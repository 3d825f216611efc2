#!/bin/bash
# Round-3 session C: half-batch pipelining of ball query + statistics pass; variants; timelines
TAG=${1:-r03c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== operator parity" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_operators_gpu.py tests/test_fullsize_gpu.py tests/test_dp_gpu.py -m gpu -q -x --timeout=600 -p no:cacheprovider > $OUT/pytest_ops.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest_ops.log | tee -a $OUT/summary.txt
for v in "" "CL3D_SPLIT=0" "CL3D_PW_QPG=1" "CL3D_PW_QPG=1 CL3D_SPLIT=0" "CL3D_PW_QPG=1 CL3D_BQ_PATH=cells" "CL3D_BQ_PATH=cells"; do
  echo "-- variant: $v" | tee -a $OUT/summary.txt
  env $v timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'])" | tee -a $OUT/summary.txt
done
echo "== timelines" | tee -a $OUT/summary.txt
for v in "" "CL3D_PW_QPG=1"; do
  rm -rf /tmp/tl
  (cd /tmp && env $v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-roofline > /dev/null 2>&1)
  echo "-- timeline: $v" | tee -a $OUT/summary.txt
  python scripts/step_timeline.py "/tmp/tl/**/tl_kernel_trace.csv" | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

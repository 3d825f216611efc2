#!/bin/bash
OUT=gpurun_out/r04e
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== suites: support pass, operators, captures, bottleneck, scene size (+ oracle rows), fp64 anchors" | tee $OUT/summary.txt
timeout 1800 python -m pytest tests/test_pwmlp_support_gpu.py tests/test_operators_gpu.py tests/test_capture_gpu.py tests/test_scene_size_gpu.py tests/test_bottleneck_gpu.py tests/test_fp64_anchor_gpu.py tests/test_fullsize_gpu.py tests/test_config2_fullsize_gpu.py -m gpu -q --timeout=900 -p no:cacheprovider 2>&1 | tail -25 | tee -a $OUT/summary.txt
echo "== step" | tee -a $OUT/summary.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null > $OUT/bench.json
python - <<PY | tee -a $OUT/summary.txt
import json
d=json.load(open("$OUT/bench.json"))
print("ms_per_step", d["ms_per_step"])
for k in d["roofline"]["step"]["kernels"]:
    print("   ", k["entry"], k["us"])
PY
echo "== timeline" | tee -a $OUT/summary.txt
(cd /tmp && rm -rf /tmp/tl && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $R/bench.py --steps 12 --warmup 4 --precondition 0 --no-cpu-baseline --no-kernel-roofline --no-step-table > /dev/null 2>&1)
python scripts/step_timeline.py "/tmp/tl/**/tl_kernel_trace.csv" | tee $OUT/step_timeline.txt | tee -a $OUT/summary.txt
echo "== two captures (DESIGN 6): which switch makes the early-stage gradients right" | tee -a $OUT/summary.txt
timeout 900 python scripts/repro_two_captures.py --config modelnet_small 2>&1 | tee $OUT/two_captures_small.jsonl | tee -a $OUT/summary.txt
timeout 1200 python scripts/repro_two_captures.py --config partnet_adaptive 2>&1 | tee $OUT/two_captures.jsonl | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

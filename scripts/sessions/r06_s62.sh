#!/bin/bash
# Round 6 session 62: the library without op_sel'd packed operands in the gather passes (pk_low, -fno-slp-vectorize): the
# interference probes of sessions 50-61 again, the two-rank bf16 replays, parity suites, and what the change costs
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s62}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== one process, bf16 contractions on a second stream: every fused operator, forward + backward" | tee $OUT/summary.txt
REPS=60 timeout 300 python scripts/micro/two_stream_survey.py 2>>$OUT/err.log | grep "bf16" | cut -c1-200 | tee -a $OUT/summary.txt
for vc in 64 144; do VC=$vc REPS=80 timeout 200 python scripts/micro/two_stream_pattern.py 2>>$OUT/err.log | grep "wrong sy" | cut -c1-330 | tee -a $OUT/summary.txt; done
echo "== beside a second PROCESS running the bf16 backbone" | tee -a $OUT/summary.txt
SKIP_ALONE=1 VICTIMS=2 LOADS=bb_bf16 REPS=100 timeout 300 python scripts/micro/pwmlp_repeat_under_load.py 2>>$OUT/err.log | cut -c1-260 | tee -a $OUT/summary.txt
echo "== two ranks on one device, bf16: replays of one step (graph, eager) and per-step norms of two runs" | tee -a $OUT/summary.txt
for mode in "" "--no-graph"; do
  CL3D_BENCH_ONE_DEVICE=1 timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --gpus 2 --warmup 1 --gemm-plans model --repeat-check 40 $mode 2>>$OUT/err.log | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('replays $mode', 'distinct late', d['distinct_late'][:6], 'early', d['distinct_early'][:6], '|', len(d['varying_parameters']), 'parameters vary')" | tee -a $OUT/summary.txt
done
for w in joined deferred; do
  CL3D_BENCH_ONE_DEVICE=1 timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --gpus 2 --steps 20 --checksums --gemm-plans model --weight-grads $w 2>>$OUT/err.log | grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$w', d.get('ms_per_step'), d.get('grad_l2'), d.get('param_l2'))" | tee -a $OUT/summary.txt
done
echo "== parity suites" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests/test_operators_gpu.py tests/test_native_gpu.py tests/test_pass_calls_gpu.py tests/test_capture_gpu.py tests/test_mfma_gemm_gpu.py tests/test_pwmlp_rows_gpu.py tests/test_bq_paths_gpu.py tests/test_fp64_anchor_gpu.py -x -q -m gpu --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== what it costs: headline (driver flags), the other operators, the ball query" | tee -a $OUT/summary.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/bench.err > $OUT/bench_driver_flags.json
python - $OUT/bench_driver_flags.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print("headline ms_per_step", d["ms_per_step"], "value", d["value"], "TRAIN us", d["roofline"]["us"], "backbone_step", (d.get("backbone_step") or {}).get("ms_per_step"))
PY
for op in pospool adaptive_weight pseudo_grid; do
  timeout 300 python bench.py --operator $op --steps 50 --backbone off --no-kernel-roofline 2>>$OUT/err.log | grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$op', d['ms_per_step'])" | tee -a $OUT/summary.txt
done
timeout 300 python bench.py --precision bf16 --steps 50 --backbone off --no-kernel-roofline 2>>$OUT/err.log | grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('pointwisemlp bf16', d['ms_per_step'])" | tee -a $OUT/summary.txt
timeout 200 python scripts/bench_bq.py 2>>$OUT/err.log | tail -3 | cut -c1-300 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

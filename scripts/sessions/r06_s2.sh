#!/bin/bash
# Round 6 session 2: in-launch K-slice sums (tests, plan sweeps both forms, config-2 backbone), one-stream capture (A/B on the
# headline and the backbone step, the formerly bad two-graph layout at 1000 replays)
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s2
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_mfma_gemm_gpu.py tests/test_bottleneck_gpu.py tests/test_config2_fullsize_gpu.py tests/test_optim_gpu.py tests/test_capture_gpu.py tests/test_pass_calls_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -12 | tee -a $OUT/summary.txt
echo "== headline, capture stream A/B (alternating)" | tee -a $OUT/summary.txt
for i in 1 2; do for cs in same separate; do
  timeout 600 python bench.py --no-cpu-baseline --no-kernel-roofline --backbone on --capture-stream $cs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cs', d['ms_per_step'], d['value'], 'backbone', d['backbone_step'].get('ms_per_step'))" | tee -a $OUT/summary.txt
done; done
echo "== config-2 backbone, bf16 and f32 (one-stream capture, in-launch slice sums)" | tee -a $OUT/summary.txt
for pr in bf16 f32; do timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision $pr 2>/dev/null | grep '^{' | tail -1 | cut -c1-400 | tee -a $OUT/summary.txt; done
for c in s3dis_pseudogrid partnet_adaptive s3dis_pospool_deep; do timeout 600 python scripts/bench_backbone.py --config $c 2>/dev/null | grep '^{' | tail -1 | cut -c1-300 | tee -a $OUT/summary.txt; done
echo "== two-graph step, formerly bad layouts on ONE stream (default now), and the old layout again" | tee -a $OUT/summary.txt
run() { local label=$1; shift
  echo "-- $label: $*" | tee -a $OUT/summary.txt
  timeout 600 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --warmup 1 --head --overlap "$@" 2>/tmp/err.txt | grep '^{' | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print({k:(l[k] if k!='distinct_early' or len(l[k])<8 else l[k][:6]+['...',len(l[k])]) for k in ('repeat_check','forks','debug','distinct_late','distinct_early')}, 'varying', len(l['varying_parameters']))" | tee -a $OUT/summary.txt || tail -3 /tmp/err.txt | tee -a $OUT/summary.txt; }
for i in 1 2 3; do run "b one stream #$i" --overlap-forks b --unsafe --repeat-check 1000; done
for i in 1 2; do run "both one stream #$i" --overlap-forks both --unsafe --repeat-check 1000; done
run "a one stream" --overlap-forks a --repeat-check 1000
run "b, warm-up stream of its own (old)" --overlap-forks b --unsafe --repeat-check 200 --debug-two-graphs other_stream
echo "== plan sweep, bf16, slices summed inside the launch / by a launch of their own" | tee -a $OUT/summary.txt
timeout 900 python scripts/micro/gemm_plan_sweep.py --run --precisions bf16 --fused-sum 1 > $OUT/gemm_plan_sweep_bf16_inlaunch.jsonl 2>$OUT/sweep.err
timeout 900 python scripts/micro/gemm_plan_sweep.py --run --precisions bf16 --fused-sum 0 > $OUT/gemm_plan_sweep_bf16_twolaunch.jsonl 2>>$OUT/sweep.err
timeout 900 python scripts/micro/gemm_plan_sweep.py --run --precisions f32 --fused-sum 1 > $OUT/gemm_plan_sweep_f32_inlaunch.jsonl 2>>$OUT/sweep.err
timeout 900 python scripts/micro/gemm_plan_sweep.py --run --precisions f32 --fused-sum 0 > $OUT/gemm_plan_sweep_f32_twolaunch.jsonl 2>>$OUT/sweep.err
wc -l $OUT/*.jsonl | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

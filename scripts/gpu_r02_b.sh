#!/bin/bash
# round 2, session B: full GPU suite + GEMM A/B + bench
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu (all)" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -25 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
echo "== A/B point GEMM" | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_point_gemm.py --sweep 2>/dev/null | tee $OUT/point_gemm.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    s = d['shape']
    print(s, ' | '.join('%s f %.1f d %.1f w %.1f' % (k, d[k]['fwd_us'], d[k]['bwd_data_us'], d[k]['bwd_weight_us']) for k in ('mfma_f32', 'mfma_bf16', 'library_f32')))
" | tee -a $OUT/summary.txt
echo "== bench (default flags)" | tee -a $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python -c "
import json
d = json.load(open('$OUT/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['ball_query_group']['frac'], d['roofline']['achieved_step'])
for r in d['roofline']['step']['kernels']: print('  %-40s %6.1f us' % (r['entry'], r['us']))
print(d['roofline']['contraction'])
print(d['cpu_baseline'])
" | tee -a $OUT/summary.txt
echo "== bench --precision bf16" | tee -a $OUT/summary.txt
timeout 600 python bench.py --precision bf16 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | tee $OUT/bench_bf16.json | cut -c1-300 | tee -a $OUT/summary.txt
echo "== backbone steps" | tee -a $OUT/summary.txt
for c in modelnet_pointwisemlp; do
  timeout 600 python scripts/bench_backbone.py --config $c 2>/dev/null | tail -1 | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

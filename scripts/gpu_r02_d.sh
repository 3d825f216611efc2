#!/bin/bash
# round 2, session D: fused bottleneck tests, GEMM re-measure, backbone steps + steady-state kernel table
TAG=${1:-r02d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest (bottleneck, gemm, operators, d2 forms)" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests/test_bottleneck_gpu.py tests/test_mfma_gemm_gpu.py tests/test_operators_gpu.py tests/test_d2_form.py tests/test_compat.py -m gpu -q --timeout=900 -p no:cacheprovider -s > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; grep -E "worst per-stage|passed|failed|^FAILED|Error:" $OUT/pytest.log | tail -25 | tee -a $OUT/summary.txt
echo "== A/B point GEMM" | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_point_gemm.py --sweep 2>/dev/null | tee $OUT/point_gemm.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    s = d['shape']
    print(s, ' | '.join('%s f %.1f d %.1f w %.1f' % (k, d[k]['fwd_us'], d[k]['bwd_data_us'], d[k]['bwd_weight_us']) for k in ('mfma_f32', 'mfma_bf16', 'library_f32')))
" | tee -a $OUT/summary.txt
echo "== bench" | tee -a $OUT/summary.txt
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python -c "
import json
d = json.load(open('$OUT/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['ball_query_group']['frac'], d['roofline']['achieved_step'])
for r in d['roofline']['step']['kernels']: print('  %-40s %6.1f us' % (r['entry'], r['us']))
c = d['roofline']['contraction']; print(c['us'], c['frac'], c['library_f32_us'], c['other_precision'])
" | tee -a $OUT/summary.txt
echo "== backbone steps" | tee -a $OUT/summary.txt
for c in modelnet_small modelnet_pointwisemlp s3dis_pseudogrid partnet_adaptive s3dis_pospool_deep; do
  timeout 600 python scripts/bench_backbone.py --config $c 2>$OUT/bb_$c.err | tail -1 | tee -a $OUT/summary.txt
done
echo "== rocprofv3 of the config-2 backbone step (40 replays)" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_bb -o bb -- python $R/scripts/bench_backbone.py --config modelnet_pointwisemlp --steps 40 > $R/$OUT/rocprof_bb.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
python scripts/kstats.py $OUT/prof_bb/bb_kernel_stats.csv 47 45 | tee -a $OUT/summary.txt
find $OUT -type f -name "*kernel_trace*" -delete 2>/dev/null
find $OUT -type f -size +3M -delete 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt

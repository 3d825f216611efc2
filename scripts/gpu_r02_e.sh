#!/bin/bash
# round 2, session E: GEMM with K slices / double-buffered LDS: tests, A/B (point GEMMs and the backbone's convolutions), backbone
TAG=${1:-r02e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4 | tee $OUT/summary.txt
echo "== pytest (gemm, bottleneck, operators)" | tee -a $OUT/summary.txt
timeout 1200 python -m pytest tests/test_mfma_gemm_gpu.py tests/test_bottleneck_gpu.py tests/test_operators_gpu.py -m gpu -q --timeout=900 -p no:cacheprovider -s > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; grep -E "worst per-stage|passed|failed|^FAILED|Error:" $OUT/pytest.log | tail -25 | tee -a $OUT/summary.txt
echo "== A/B point GEMM" | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_point_gemm.py --sweep --reps 30 2>/dev/null | tee $OUT/point_gemm.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    s = d['shape']
    print(s, ' | '.join('%s f %.1f d %.1f w %.1f' % (k, d[k]['fwd_us'], d[k]['bwd_data_us'], d[k]['bwd_weight_us']) for k in ('mfma_f32', 'mfma_bf16', 'library_f32')))
" | tee -a $OUT/summary.txt
echo "== convolutions of config 2: engine f32 [fwd, dx, dW] / bf16 / library [fwd, bwd]" | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_point_gemm.py --convs --reps 20 2>/dev/null | tee $OUT/convs.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); c = d['conv']
    print('%4d->%4d N=%4d  f32 %s  bf16 %s  lib %s' % (c['C'], c['Co'], c['N'], d['f32'], d['bf16'], d['library_f32']))
" | tee -a $OUT/summary.txt
echo "== backbone steps" | tee -a $OUT/summary.txt
for c in modelnet_pointwisemlp s3dis_pseudogrid; do
  timeout 600 python scripts/bench_backbone.py --config $c 2>$OUT/bb_$c.err | tail -1 | tee -a $OUT/summary.txt
  timeout 600 python scripts/bench_backbone.py --config $c --impl grouped 2>>$OUT/bb_$c.err | tail -1 | tee -a $OUT/summary.txt
done
echo "== rocprofv3 of the config-2 backbone step (40 replays)" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_bb -o bb -- python $R/scripts/bench_backbone.py --config modelnet_pointwisemlp --steps 40 > $R/$OUT/rocprof_bb.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
python scripts/kstats.py $OUT/prof_bb/bb_kernel_stats.csv 47 40 | tee -a $OUT/summary.txt
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4 | tee -a $OUT/summary.txt
find $OUT -type f -name "*kernel_trace*" -delete 2>/dev/null
find $OUT -type f -size +3M -delete 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt

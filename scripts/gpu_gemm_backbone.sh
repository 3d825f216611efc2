#!/bin/bash
TAG=${1:-r02f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest (gemm, bottleneck)" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests/test_mfma_gemm_gpu.py tests/test_bottleneck_gpu.py tests/test_operators_gpu.py -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; grep -E "passed|failed|^FAILED|Error:" $OUT/pytest.log | tail -10 | tee -a $OUT/summary.txt
echo "== raw kernel rate: 4096^3 as a 1x1 convolution (f32 / bf16 [fwd, dx, dW] us; 137.4 GFLOP each)" | tee -a $OUT/summary.txt
timeout 300 python -c "
import sys; sys.path.insert(0, 'scripts')
import bench_point_gemm as b, json
r = b.measure_conv(1, 4096, 4096, 4096, reps=5)
print(r['f32'], r['bf16'], r['library_f32'], 'TF f32 fwd', 137.4e9 / r['f32'][0] / 1e6, 'bf16 fwd', 137.4e9 / r['bf16'][0] / 1e6)
" 2>/dev/null | tee -a $OUT/summary.txt
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
echo "== backbone config 2: engine convs f32 / bf16 / library convs (+ engine BatchNorm passes) / grouped is the reference dataflow" | tee -a $OUT/summary.txt
CL3D_BLOCK=modules timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp 2>$OUT/bb.err | tail -1 | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp 2>>$OUT/bb.err | tail -1 | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>>$OUT/bb.err | tail -1 | tee -a $OUT/summary.txt
CL3D_CONV=library timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp 2>>$OUT/bb.err | tail -1 | tee -a $OUT/summary.txt
echo "== rocprofv3 of the config-2 backbone step (40 replays)" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_bb -o bb -- python $R/scripts/bench_backbone.py --config modelnet_pointwisemlp --steps 40 > $R/$OUT/rocprof_bb.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
python scripts/kstats.py $OUT/prof_bb/bb_kernel_stats.csv 47 30 | tee -a $OUT/summary.txt
find $OUT -type f -name "*kernel_trace*" -delete 2>/dev/null
find $OUT -type f -size +3M -delete 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt

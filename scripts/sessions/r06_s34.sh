#!/bin/bash
# Round 6 session 34: deferred weight gradients (closerlook3d_amd.deferred_weight_gradients: d W of every contraction stays on the side
# stream beside the rest of the backward pass, one join in front of the optimizer) -- the new capture tests, then every backbone config
# joined / deferred under the runtime's default graph layout and under DEBUG_HIP_FORCE_GRAPH_QUEUES=3, alternating
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s34}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
line() { grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(sys.argv[1], d.get('ms_per_step'))" "$1"; }
echo "== pytest" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_capture_gpu.py tests/test_bottleneck_gpu.py tests/test_dp_gpu.py -x -q -m gpu --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -15 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
for q in default 3; do
  if [ $q = default ]; then unset DEBUG_HIP_FORCE_GRAPH_QUEUES; else export DEBUG_HIP_FORCE_GRAPH_QUEUES=$q; fi
  for cfg in "modelnet_pointwisemlp --precision bf16" "modelnet_pointwisemlp"; do
    echo "== backbone $cfg, queues=$q, joined / deferred" | tee -a $OUT/summary.txt
    for i in 1 2 3; do
      timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 2>>$OUT/err.log | line joined | tee -a $OUT/summary.txt
      timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 --weight-grads deferred 2>>$OUT/err.log | line deferred | tee -a $OUT/summary.txt
    done
  done
  for cfg in s3dis_pseudogrid partnet_adaptive s3dis_pospool_deep; do
    echo "== backbone $cfg, queues=$q, joined / deferred" | tee -a $OUT/summary.txt
    for i in 1 2; do
      timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 2>>$OUT/err.log | line joined | tee -a $OUT/summary.txt
      timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 --weight-grads deferred 2>>$OUT/err.log | line deferred | tee -a $OUT/summary.txt
    done
  done
done
unset DEBUG_HIP_FORCE_GRAPH_QUEUES
echo "== checksums, config 2 bf16: joined / deferred (gradient and parameter norms after the same steps)" | tee -a $OUT/summary.txt
for w in joined deferred; do
  timeout 400 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 10 --checksums --gemm-plans model --weight-grads $w 2>>$OUT/err.log | grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(sys.argv[1], {k:v for k,v in d.items() if 'norm' in k or 'checksum' in k})" $w | tee -a $OUT/summary.txt
done
tail -5 $OUT/err.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== done" | tee -a $OUT/summary.txt

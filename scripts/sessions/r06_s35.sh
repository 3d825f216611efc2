#!/bin/bash
# Round 6 session 35: deferred weight gradients, second form -- session 34 read them SLOWER than the layer-by-layer joins (config 2 bf16
# 5.65 -> 5.72-5.80 ms, 5.415 -> 5.61 under three graph queues): with the weight gradient captured first, the chain the backward pass
# waits for was every fork's SECOND dependent and changed queue there.  Now the caller's piece is captured first.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s35}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
line() { grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(sys.argv[1], d.get('ms_per_step'), d.get('grad_l2',''), d.get('param_l2',''), d.get('peak_mem_GB',''))" "$1"; }
echo "== pytest" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_capture_gpu.py -x -q -m gpu --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
for q in default 3; do
  if [ $q = default ]; then unset DEBUG_HIP_FORCE_GRAPH_QUEUES; else export DEBUG_HIP_FORCE_GRAPH_QUEUES=$q; fi
  for cfg in "modelnet_pointwisemlp --precision bf16" "modelnet_pointwisemlp" "partnet_adaptive"; do
    echo "== backbone $cfg, queues=$q, joined / deferred" | tee -a $OUT/summary.txt
    for i in 1 2 3; do
      timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 2>>$OUT/err.log | line joined | tee -a $OUT/summary.txt
      timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 --weight-grads deferred 2>>$OUT/err.log | line deferred | tee -a $OUT/summary.txt
    done
  done
done
unset DEBUG_HIP_FORCE_GRAPH_QUEUES
echo "== checksums, config 2 bf16: joined / deferred (gradient and parameter norms after the same steps)" | tee -a $OUT/summary.txt
for w in joined deferred; do
  timeout 400 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 10 --checksums --gemm-plans model --weight-grads $w 2>>$OUT/err.log | line $w | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

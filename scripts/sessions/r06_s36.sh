#!/bin/bash
# Round 6 session 36: deferred weight gradients as the benches' default (session 35: config 2 bf16 5.60 -> 5.09 ms, 4.89 under three
# graph queues): the suites that touch it (capture, two ranks on one device, bottlenecks, operators), the driver's line with its
# backbone leg in a child process, every backbone config, and where the remaining idle time of the config-2 step sits
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${1:-r06_s36}
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
line() { grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(sys.argv[1], d.get('ms_per_step'), d.get('weight_grads',''), d.get('graph_queues',''), d.get('peak_mem_GB',''))" "$1"; }
echo "== pytest" | tee $OUT/summary.txt
timeout 2000 python -m pytest tests/test_capture_gpu.py tests/test_dp_gpu.py tests/test_bottleneck_gpu.py tests/test_operators_gpu.py tests/test_config2_fullsize_gpu.py -x -q -m gpu --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -6 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== the driver's command" | tee -a $OUT/summary.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/bench.err > $OUT/bench_driver_flags.json
python - $OUT/bench_driver_flags.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"])
print("backbone_step", d.get("backbone_step"))
PY
tail -3 $OUT/bench.err | cut -c1-300 | tee -a $OUT/summary.txt
for q in default 3; do
  if [ $q = default ]; then unset DEBUG_HIP_FORCE_GRAPH_QUEUES; else export DEBUG_HIP_FORCE_GRAPH_QUEUES=$q; fi
  echo "== backbones, queues=$q: joined / deferred" | tee -a $OUT/summary.txt
  for cfg in "modelnet_pointwisemlp --precision bf16" s3dis_pseudogrid partnet_adaptive s3dis_pospool_deep "modelnet_small"; do
    for i in 1 2; do
      timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 --weight-grads joined 2>>$OUT/err.log | line "$cfg joined" | tee -a $OUT/summary.txt
      timeout 400 python scripts/bench_backbone.py --config $cfg --steps 30 2>>$OUT/err.log | line "$cfg deferred" | tee -a $OUT/summary.txt
    done
  done
done
echo "== two ranks on one device (gloo), config 2 bf16: joined / deferred" | tee -a $OUT/summary.txt
for w in joined deferred; do
  CL3D_BENCH_ONE_DEVICE=1 timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --gpus 2 --steps 20 --checksums --gemm-plans model --weight-grads $w 2>>$OUT/err.log | grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(sys.argv[1], d.get('ms_per_step'), d.get('grad_l2'), d.get('param_l2'))" $w | tee -a $OUT/summary.txt
done
echo "== idle time of the config-2 step (bf16, deferred, three queues)" | tee -a $OUT/summary.txt
export DEBUG_HIP_FORCE_GRAPH_QUEUES=3
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/prof_bb -o bb -- python $R/scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 150 > $R/$OUT/rocprof_bb.log 2>&1)
T=$(find $OUT/prof_bb -name "bb_kernel_trace.csv" | head -1)
python scripts/graph_idle.py "$T" | tee -a $OUT/summary.txt
rm -rf $OUT/prof_bb
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# Round 6 session 17: measured GEMM plans as the benches' default: tests, the driver's bench line, config 2 both ways.
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s17
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_mfma_gemm_gpu.py tests/test_bottleneck_gpu.py tests/test_config2_fullsize_gpu.py tests/test_abi_host_gpu.py -q -m gpu --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== python bench.py (driver flags)" | tee -a $OUT/summary.txt
( time timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err ) 2>&1 | grep real | tee -a $OUT/summary.txt
python - <<'PY' | tee -a $OUT/summary.txt
import json
d = json.loads(open("gpurun_out/r06_s17/bench_line.json").read().strip().splitlines()[-1])
print("headline", d["ms_per_step"], d["value"])
print("backbone_step", json.dumps(d.get("backbone_step"))[:700])
PY
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d.get('ms_per_step'), d.get('gemm_plans_measured'), d.get('gemm_plans_changed'))" "$1"; }
echo "== config 2 bf16 with the seg head / two ranks on one device are not run here; plain config 2:" | tee -a $OUT/summary.txt
for i in 1 2; do
  timeout 400 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --gemm-plans model 2>/dev/null | line model | tee -a $OUT/summary.txt
  timeout 400 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | line measured | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

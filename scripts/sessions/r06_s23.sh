#!/bin/bash
# Round 6 session 23: S_a = sum y rel_a out of the TRAIN walk (terms 1 + 2 per query in the statistics pass, term 3 per support
# point in the support-major pass, ABI 6): parity suites, then the headline step against the tree at HEAD (scripts/micro/var/head_tree)
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s23
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
echo "== pytest" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests/test_pwmlp_support_gpu.py tests/test_operators_gpu.py tests/test_fp64_anchor_gpu.py tests/test_pass_calls_gpu.py tests/test_abi_host_gpu.py tests/test_capture_gpu.py tests/test_bottleneck_gpu.py tests/test_pwmlp_rows_gpu.py -x -q -m gpu --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -15 $OUT/pytest.log | cut -c1-400 | tee -a $OUT/summary.txt
line() { grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); st=r.get('step',{})
print(sys.argv[1], d.get('ms_per_step'), round(d.get('value',0)/1e6,1), ' '.join('%s=%.1f'%(k['entry'].replace('cl3d_pwmlp_',''),k['us']) for k in st.get('kernels',[])[:8]))" "$1"; }
echo "== headline, new / head alternating (step table in the first pair)" | tee -a $OUT/summary.txt
(cd $R && timeout 300 python bench.py --steps 100 --no-cpu-baseline --backbone off 2>/dev/null | tee $OUT/bench_new.json | line new | tee -a $OUT/summary.txt)
(cd $R/scripts/micro/var/head_tree && timeout 300 python bench.py --steps 100 --no-cpu-baseline --backbone off 2>/dev/null | tee $R/$OUT/bench_head.json | line head | tee -a $R/$OUT/summary.txt)
for i in 1 2 3; do
  (cd $R && timeout 300 python bench.py --steps 100 --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line new | tee -a $OUT/summary.txt)
  (cd $R/scripts/micro/var/head_tree && timeout 300 python bench.py --steps 100 --no-cpu-baseline --backbone off --no-step-table --no-kernel-roofline 2>/dev/null | line head | tee -a $R/$OUT/summary.txt)
done
echo "== config 2 backbone bf16, new / head" | tee -a $OUT/summary.txt
for i in 1 2; do
  (cd $R && timeout 400 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | line new | tee -a $OUT/summary.txt)
  (cd $R/scripts/micro/var/head_tree && timeout 400 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | line head | tee -a $R/$OUT/summary.txt)
done
echo "== done" | tee -a $OUT/summary.txt

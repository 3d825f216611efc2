#!/bin/bash
OUT=gpurun_out/r05d
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bm() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms_per_step', d['ms_per_step'], d['config']['launch'])"; }
for i in 1 2; do
  for v in product stats apply; do
    CL3D_CSR_AFTER=$v timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "csr after $v" | tee -a $OUT/summary.txt
  done
done
for v in stats apply; do
(cd /tmp && CL3D_CSR_AFTER=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof$v -o bench -- python $R/bench.py --no-cpu-baseline --no-kernel-roofline --precondition 0 --steps 50 > $R/$OUT/rocprof.log 2>&1)
python scripts/step_timeline.py "$OUT/prof$v/**/bench_kernel_trace.csv" | tee $OUT/step_timeline_$v.txt | tee -a $OUT/summary.txt
done

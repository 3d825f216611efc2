#!/bin/bash
OUT=gpurun_out/r05m
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bm() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms_per_step', d['ms_per_step'], d['config']['launch'])"; }
for rep in 1 2; do
for v in shipped csr_wpb2 csr_wpb1; do
  if [ $v = shipped ]; then L=""; else L="CL3D_LIB=$R/scripts/micro/var/libcl3d_$v.so"; fi
  env $L timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "$v" | tee -a $OUT/summary.txt
done
done
for v in csr_wpb2 csr_wpb1; do
(cd /tmp && CL3D_LIB=$R/scripts/micro/var/libcl3d_$v.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof$v -o bench -- python $R/bench.py --no-cpu-baseline --no-kernel-roofline --precondition 0 --steps 50 > $R/$OUT/rocprof.log 2>&1)
python scripts/step_timeline.py "$OUT/prof$v/**/bench_kernel_trace.csv" | tee $OUT/step_timeline_$v.txt | grep "csr\|query\|rows_kernel<0>\|step of" | tee -a $OUT/summary.txt
done
CL3D_LIB=$R/scripts/micro/var/libcl3d_csr_wpb1.so timeout 300 python -m pytest tests/test_operators_gpu.py -m gpu -q -x -p no:cacheprovider -k "inverse_index" 2>&1 | tail -2 | tee -a $OUT/summary.txt

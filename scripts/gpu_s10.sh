#!/bin/bash
OUT=gpurun_out/r05l
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bm() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms_per_step', d['ms_per_step'], d['config']['launch'])"; }
for rep in 1 2; do
for v in shipped csr_prio1 csr_prio2 csr_prio3; do
  if [ $v = shipped ]; then L=""; else L="CL3D_LIB=$R/scripts/micro/var/libcl3d_$v.so"; fi
  env $L timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | bm "$v" | tee -a $OUT/summary.txt
done
done
for v in csr_prio3; do
(cd /tmp && CL3D_LIB=$R/scripts/micro/var/libcl3d_$v.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof$v -o bench -- python $R/bench.py --no-cpu-baseline --no-kernel-roofline --precondition 0 --steps 50 > $R/$OUT/rocprof.log 2>&1)
python scripts/step_timeline.py "$OUT/prof$v/**/bench_kernel_trace.csv" | tee $OUT/step_timeline_$v.txt | tee -a $OUT/summary.txt
done
for c in s3dis_pseudogrid partnet_adaptive; do
for v in shipped csr_prio3; do
  if [ $v = shipped ]; then L=""; else L="CL3D_LIB=$R/scripts/micro/var/libcl3d_$v.so"; fi
  env $L timeout 600 python scripts/bench_backbone.py --config $c 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', '$v', d['ms_per_step'])" | tee -a $OUT/summary.txt
done
done

#!/bin/bash
# rocprofv3 kernel summary of the bench step for each operator.  Usage: bash scripts/prof_ops.sh <tag> [operators...]
TAG=${1:-ops}; shift
OPS=${@:-pointwisemlp pospool adaptive_weight pseudo_grid}
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
for op in $OPS; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$op -o bench -- \
     python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-roofline --operator $op > $O/rocprof_$op.log 2>&1)
  echo "== $op: $(tail -1 $O/rocprof_$op.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["launch"])' 2>/dev/null)"
  python $GRAFT_REPO_ROOT/scripts/kstats.py $O/prof_$op/bench_kernel_stats.csv 49 ${TOPN:-16}
  find $O/prof_$op -type f ! -name "*stats*" -size +1M -delete 2>/dev/null
done

#!/bin/bash
# Round 6 session 20c: the pooling kernels on the wide maps (shipped build) / on rounds 1-5's (variant maxpool_narrow)
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s20
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
V=$R/scripts/micro/var/libcl3d_maxpool_narrow.so
line() { grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d.get('ms_per_step'), d.get('value'))" "$1"; }
echo "== (c) pooling kernels: wide maps / narrow maps" | tee $OUT/summary_c.txt
for cfg in "s3dis_pospool_deep" "s3dis_pseudogrid" "partnet_adaptive" "modelnet_pointwisemlp --precision bf16"; do
  echo "== backbone $cfg" | tee -a $OUT/summary_c.txt
  for i in 1 2; do
    timeout 400 python scripts/bench_backbone.py --config $cfg 2>/dev/null | line wide | tee -a $OUT/summary_c.txt
    CL3D_LIB=$V timeout 400 python scripts/bench_backbone.py --config $cfg 2>/dev/null | line narrow | tee -a $OUT/summary_c.txt
  done
done
echo "== done" | tee -a $OUT/summary_c.txt

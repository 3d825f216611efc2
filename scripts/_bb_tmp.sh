for c in "" "--precision bf16"; do
  timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp $c 2>&1 | tail -4 | cut -c1-400
done
CL3D_BLOCK=modules timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp 2>&1 | tail -2 | cut -c1-400

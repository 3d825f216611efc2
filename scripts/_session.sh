timeout 900 python -m pytest tests/test_pwmlp_rows_gpu.py tests/test_operators_gpu.py tests/test_config2_fullsize_gpu.py tests/test_bottleneck_gpu.py tests/test_pass_calls_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | tail -1 | cut -c150-260; done
timeout 600 python bench.py --no-cpu-baseline --no-kernel-roofline --backbone off 2>/dev/null | cut -c1-260

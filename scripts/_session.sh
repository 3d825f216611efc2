OUT=gpurun_out/r05s5; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pwmlp_rows_gpu.py tests/test_pass_calls_gpu.py tests/test_operators_gpu.py tests/test_fullsize_gpu.py tests/test_abi_host_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --backbone off > $OUT/bench$i.json 2> $OUT/bench.err; cut -c1-330 $OUT/bench$i.json; done
timeout 600 python bench.py --no-graph --no-cpu-baseline --no-kernel-roofline --backbone off 2>/dev/null | cut -c1-300

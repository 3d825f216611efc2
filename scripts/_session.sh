OUT=gpurun_out/r05s6; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_native_gpu.py tests/test_ref_pin_gpu.py tests/test_scene_size_gpu.py -x -q -p no:cacheprovider -k "subsampl or grid" 2>&1 | tail -3
for i in 1 2; do
  (cd _wt_old && timeout 600 python bench.py --no-graph --no-cpu-baseline --no-kernel-roofline --backbone off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OLD eager', d['ms_per_step'])")
  timeout 600 python bench.py --no-graph --no-cpu-baseline --no-kernel-roofline --backbone off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NEW eager', d['ms_per_step'])"
  (cd _wt_old && timeout 600 python bench.py --no-cpu-baseline --no-kernel-roofline --backbone off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OLD graph', d['ms_per_step'])")
  timeout 600 python bench.py --no-cpu-baseline --no-kernel-roofline --backbone off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NEW graph', d['ms_per_step'])"
done
(cd _wt_old && timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | tail -1 | cut -c150-260)
timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | tail -1 | cut -c150-260

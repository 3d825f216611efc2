python scripts/micro/grid_subsample_phases.py 2>&1
for i in 1 2 3; do timeout 900 python -m pytest tests/test_native_gpu.py tests/test_ref_pin_gpu.py tests/test_scene_size_gpu.py -x -q -p no:cacheprovider --tb=short 2>&1 | tail -12; done

#!/bin/bash
# segmentation decoders: parity (seg-head fixtures, reference configs) and the three segmentation backbone configs
TAG=${1:-dec}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_operators_gpu.py tests/test_bottleneck_gpu.py tests/test_scene_size_gpu.py tests/test_dp_gpu.py -m gpu -q --timeout=300 -p no:cacheprovider -k "resnet or seg or reference_configs or scene or dp or bottleneck" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee $OUT/summary.txt; grep -E "passed|failed|^FAILED|Error:|assert " $OUT/pytest.log | tail -8 | tee -a $OUT/summary.txt
for c in s3dis_pseudogrid partnet_adaptive s3dis_pospool_deep; do
  timeout 300 python scripts/bench_backbone.py --config $c --head 2>/dev/null | tail -1 | cut -c1-260 | tee -a $OUT/summary.txt
  CL3D_DECODE=cat timeout 300 python scripts/bench_backbone.py --config $c --head 2>/dev/null | tail -1 | cut -c1-240 | sed 's/^/concatenating decoder: /' | tee -a $OUT/summary.txt
done
echo "== done" | tee -a $OUT/summary.txt

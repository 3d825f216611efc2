#!/bin/bash
# Round 6 session 6: after the CSR / sort tuning: tests of the touched files, scene configs, config-5 kernel table
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s6
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest" | tee $OUT/summary.txt
timeout 2400 python -m pytest tests/test_pass_calls_gpu.py tests/test_abi_host_gpu.py tests/test_operators_gpu.py tests/test_scene_size_gpu.py tests/test_native_gpu.py tests/test_ref_pin_gpu.py tests/test_d2_form.py tests/test_mfma_gemm_gpu.py tests/test_bq_tune.py tests/test_dp_gpu.py tests/test_sphere_crop.py tests/test_dataset_grid.py -q -m gpu --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -12 $OUT/pytest.log | cut -c1-400 | tee -a $OUT/summary.txt
echo "== backbones" | tee -a $OUT/summary.txt
for i in 1 2; do for c in s3dis_pseudogrid s3dis_pospool_deep; do timeout 600 python scripts/bench_backbone.py --config $c 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config'], d['ms_per_step'])" | tee -a $OUT/summary.txt; done; done
echo "== kernel table of config 5: the sorts" | tee -a $OUT/summary.txt
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c5 -o bb -- python $R/scripts/bench_backbone.py --config s3dis_pospool_deep --steps 20 > $R/$OUT/rocprof_c5.log 2>&1)
python scripts/kstats.py $(find $OUT/prof_c5 -name "bb_kernel_stats.csv" | head -1) 27 60 | grep -i "csr\|grid_s\|sort\|total" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c3 -o bb -- python $R/scripts/bench_backbone.py --config s3dis_pseudogrid --steps 20 > $R/$OUT/rocprof_c3.log 2>&1)
python scripts/kstats.py $(find $OUT/prof_c3 -name "bb_kernel_stats.csv" | head -1) 27 60 | grep -i "csr\|grid_s\|sort\|total" | tee -a $OUT/summary.txt
find $OUT -name "*kernel_trace*" -delete 2>/dev/null; find $OUT -type f -size +3M -delete 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt

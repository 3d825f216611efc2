#!/bin/bash
# Copy the summaries of one `bash scripts/gpu_check.sh <tag>` session from gpurun_out/<tag>/ into profiles/<round>/
# under the names profiles/README.md lists.   bash scripts/collect_profiles.sh r05z r05
TAG=$1; ROUND=${2:-r06}
S=gpurun_out/$TAG; D=profiles/$ROUND
mkdir -p $D
cp $S/summary.txt $D/gpu_check_summary.txt
cp $S/bench.json $D/bench_pointwisemlp.json
cp $S/bench_driver_flags.json $D/bench_pointwisemlp_driver_flags.json
cp $S/bench_bf16.json $D/bench_pointwisemlp_bf16.json
cp $S/bench_eager.json $D/bench_pointwisemlp_eager.json
cp $(find $S/prof -name "bench_kernel_stats.csv" | head -1) $D/bench_pointwisemlp_kernel_stats.csv
for f in step_counters.json step_timeline.txt pmc_traffic.json point_gemm.jsonl convs.jsonl bench_bq.jsonl \
         bench_pospool.json bench_adaptive_weight.json bench_pseudo_grid.json bench_dataset_grid.json bench_voting.json \
         bench_sphere_crop.json eager_host.txt graph_queues.txt bench_two_ranks_one_device.json; do
  cp $S/$f $D/$f 2>/dev/null || echo "missing $f"
done
cp $S/backbone_steady_state.txt $D/backbone_modelnet_pointwisemlp_bf16_steady_state.txt
cp $(find $S/prof_bb -name "bb_kernel_stats.csv" | head -1) $D/backbone_modelnet_pointwisemlp_bf16_kernel_stats.csv
cp $(find $S/pmc_fetch -name "*counter_collection.csv" | head -1) $D/pmc_fetch_size_counter_collection.csv
cp $(find $S/pmc_write -name "*counter_collection.csv" | head -1) $D/pmc_write_size_counter_collection.csv
ls -la $D | wc -l

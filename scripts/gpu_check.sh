#!/bin/bash
# One GPU-box session: parity tests, reference pin, bench (+ rocprof summary of the same command), PMC counters of the
# timed step and of the ball_query+group boundary, the contraction, the other operators, the backbone configs, the
# data-parallel stand-ins and the round's micro-benchmarks.
# Usage (from the repo root on the GPU box): bash scripts/gpu_check.sh [tag]      (then scripts/collect_profiles.sh)
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== smoke" | tee $OUT/summary.txt
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
echo "== pytest -m gpu (engine vs oracle, golden fixtures, reference pin, full-size and scene-size properties, data parallel)" | tee -a $OUT/summary.txt
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -8 $OUT/pytest_gpu.log | cut -c1-300 | tee -a $OUT/summary.txt
echo "== bench (the driver's command, default flags)" | tee -a $OUT/summary.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err; echo "bench (driver's flags) rc=$?" | tee -a $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2>> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench.json | cut -c1-3000 | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
for f in ("bench_driver_flags.json", "bench.json"):
    d = json.load(open("$OUT/" + f)); r = d["roofline"]
    print(f, "ms_per_step", d["ms_per_step"], d["config"]["launch"], "| top", r["kernel"], r["us"], "us frac", r["frac"], "| boundary", r["boundary"]["ball_query_group"]["frac"], "min", r["boundary"]["ball_query_group"]["frac_min"], "| achieved_step", r["achieved_step"]["frac"], "| cpu", d["cpu_baseline"]["all_cores"]["value"], d["cpu_baseline"]["one_thread"]["value"])
    print("   contraction", {k: r["contraction"].get(k) for k in ("us", "fwd_us", "bwd_both_us", "separate_products_us", "frac", "hbm_frac")})
    print("   backbone_step", d.get("backbone_step"))
PY
echo "== bench --precision bf16 (config 2's arithmetic)" | tee -a $OUT/summary.txt
timeout 900 python bench.py --precision bf16 --no-cpu-baseline --backbone off 2>/dev/null | tee $OUT/bench_bf16.json | cut -c1-300 | tee -a $OUT/summary.txt
echo "== bench, eager launches" | tee -a $OUT/summary.txt
timeout 900 python bench.py --no-graph --no-cpu-baseline --no-kernel-roofline 2>/dev/null | tee $OUT/bench_eager.json | cut -c1-260 | tee -a $OUT/summary.txt
echo "== ball query: LDS-resident kernel vs cell grid through HBM (same op, pinned path)" | tee -a $OUT/summary.txt
for p in tile cells; do CL3D_BQ_PATH=$p timeout 120 python scripts/bench_bq.py | tee -a $OUT/bench_bq.jsonl | tee -a $OUT/summary.txt; done
for p in tile cells; do CL3D_BQ_PATH=$p timeout 120 python scripts/bench_bq.py --mult 4.0 | tee -a $OUT/bench_bq.jsonl | tee -a $OUT/summary.txt; done
echo "== how the runtime lays a captured step out on its graph queues (timed spin kernels, every capture order)" | tee -a $OUT/summary.txt
[ -x scripts/micro/graph_queues ] || (cd scripts/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o graph_queues graph_queues.hip > /dev/null 2>&1)
(cd scripts/micro && timeout 60 ./graph_queues) > $OUT/graph_queues.txt 2>&1; grep "==\|period" $OUT/graph_queues.txt | tee -a $OUT/summary.txt
echo "== eager step: host enqueue time against drained time (one C-ABI call per pass)" | tee -a $OUT/summary.txt
timeout 300 python scripts/micro/eager_host.py 2>/dev/null | head -12 | tee $OUT/eager_host.txt | head -3 | tee -a $OUT/summary.txt
echo "== rocprofv3 kernel trace of the same bench command" | tee -a $OUT/summary.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --no-cpu-baseline --backbone off --precondition 0 > $R/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
python scripts/kstats.py $(find $OUT/prof -name "bench_kernel_stats.csv" | head -1) 100 40 | tee -a $OUT/summary.txt
python scripts/step_timeline.py "$OUT/prof/**/bench_kernel_trace.csv" | tee $OUT/step_timeline.txt | tee -a $OUT/summary.txt
echo "== PMC counters of the timed step (separate passes)" | tee -a $OUT/summary.txt
n=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" \
            "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES SQ_BUSY_CYCLES" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$((n+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$OUT/step_pmc$n -o pmc -- python $R/bench.py --steps 6 --warmup 2 --precondition 0 --no-cpu-baseline --no-kernel-roofline > $R/$OUT/step_pmc$n.log 2>&1)
done
python scripts/step_counters.py $OUT/step_pmc1 $OUT/step_pmc2 $OUT/step_pmc3 $OUT/step_pmc4 $OUT/step_pmc5 $OUT/step_pmc6 > $OUT/step_counters.json 2>> $OUT/summary.txt
head -c 1500 $OUT/step_counters.json | tee -a $OUT/summary.txt
echo "== PMC traffic of the ball_query+group kernels (separate passes)" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_fetch -o pmc -- python $R/scripts/pmc_kernels.py > $R/$OUT/pmc_fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_write -o pmc -- python $R/scripts/pmc_kernels.py > $R/$OUT/pmc_write.log 2>&1)
python scripts/pmc_kernels.py --parse $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json 2>> $OUT/summary.txt
head -c 1200 $OUT/pmc_traffic.json | tee -a $OUT/summary.txt
echo "== contraction (engine MFMA f32 / bf16 vs the vendor library; both gradients in one call)" | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_point_gemm.py --sweep 2>/dev/null | tee $OUT/point_gemm.jsonl | cut -c1-800 | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_point_gemm.py --convs --reps 20 2>/dev/null > $OUT/convs.jsonl
echo "== other operators (bench.py --operator)" | tee -a $OUT/summary.txt
for op in pospool adaptive_weight pseudo_grid; do
  # (with the step table: the PMC columns come from profiles/rNN/step_counters_$op.json, scripts/sessions/r06_s4.sh collects them)
  timeout 600 python bench.py --operator $op --no-cpu-baseline --backbone off 2>/dev/null | tee $OUT/bench_$op.json | cut -c1-330 | tee -a $OUT/summary.txt
done
echo "== backbone steps (scripts/bench_backbone.py); configs 3 / 4 / 5 also layer by layer (f1 off)" | tee -a $OUT/summary.txt
for c in modelnet_small modelnet_pointwisemlp s3dis_pseudogrid partnet_adaptive s3dis_pospool_deep; do
  timeout 600 python scripts/bench_backbone.py --config $c 2>/dev/null | tail -1 | cut -c1-330 | tee -a $OUT/summary.txt
done
timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | tail -1 | cut -c1-330 | tee -a $OUT/summary.txt
for c in s3dis_pseudogrid partnet_adaptive s3dis_pospool_deep; do
  timeout 600 python scripts/bench_backbone.py --config $c --layerwise 2>/dev/null | tail -1 | cut -c1-330 | sed 's/^/layer by layer: /' | tee -a $OUT/summary.txt
done
echo "== segmentation configs with the scene-segmentation head in the step" | tee -a $OUT/summary.txt
for c in s3dis_pseudogrid partnet_adaptive s3dis_pospool_deep; do
  timeout 600 python scripts/bench_backbone.py --config $c --head 2>/dev/null | tail -1 | cut -c1-330 | tee -a $OUT/summary.txt
done
echo "== steady-state kernel table of the config-2 backbone step (bf16)" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_bb -o bb -- python $R/scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 40 > $R/$OUT/rocprof_bb.log 2>&1)
python scripts/kstats.py $(find $OUT/prof_bb -name "bb_kernel_stats.csv" | head -1) 47 50 | tee $OUT/backbone_steady_state.txt | head -30 | tee -a $OUT/summary.txt
echo "== data parallel on one device (2 ranks over gloo): backbone, flat exchange and two-graph overlapped exchange; bench.py with its backbone step" | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_backbone.py --gpus 2 --config partnet_adaptive 2>/dev/null | grep '^{' | tail -1 | cut -c1-500 | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_backbone.py --gpus 2 --config partnet_adaptive --overlap 2>/dev/null | grep '^{' | tail -1 | cut -c1-500 | tee -a $OUT/summary.txt
timeout 600 python bench.py --gpus 2 --no-cpu-baseline --no-kernel-roofline --backbone on 2>/dev/null | grep '^{' | tail -1 | tee $OUT/bench_two_ranks_one_device.json | cut -c1-1500 | tee -a $OUT/summary.txt
echo "== two-graph step, 200 replays without the update: distinct bit patterns of the exchanged gradients (DESIGN 6)" | tee -a $OUT/summary.txt
for cfg in "" "--overlap --overlap-forks none" "--overlap --overlap-forks a" "--overlap --overlap-forks b" "--overlap --overlap-forks both" "--overlap --overlap-forks b --debug-two-graphs other_stream --unsafe"; do
  timeout 300 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --warmup 1 --head $cfg --repeat-check 200 2>/dev/null | grep repeat_check | python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['varying_parameters']
print('[$cfg]', 'late', d['distinct_late'][:4], 'early', d['distinct_early'][:4], 'varying parameters', len(v))" | tee -a $OUT/two_graph_repeat_check.txt | tee -a $OUT/summary.txt
done
timeout 60 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --head --overlap --overlap-forks b --debug-two-graphs other_stream 2>&1 | tail -2 | cut -c1-400 | sed 's/^/without --unsafe: /' | tee -a $OUT/two_graph_repeat_check.txt | tee -a $OUT/summary.txt
echo "== dataset-side grid subsampling, voting, sphere crops" | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_dataset_grid.py 2>/dev/null | tee $OUT/bench_dataset_grid.json | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_voting.py 2>/dev/null | tee $OUT/bench_voting.json | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_sphere_crop.py 2>/dev/null | tee $OUT/bench_sphere_crop.json | tee -a $OUT/summary.txt
# closing cleanup: the rocprof traces of the PMC / kernel-trace passes stay out of what is merged back (the
# summaries above are what is kept), and the marker tells a finished session from a truncated one
find $OUT -name "*kernel_trace*" -delete 2>/dev/null
find $OUT -type f -size +3M -delete 2>/dev/null
echo "== done $(date -u +%H:%M:%S)" | tee -a $OUT/summary.txt

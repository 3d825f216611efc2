#!/bin/bash
# One GPU-box session: parity tests, reference pin, bench (+ rocprof summary of the same command), PMC counters of the
# timed step and of the ball_query+group boundary, contraction A/B, the other operators and the backbone configs.
# Usage (from the repo root on the GPU box): bash scripts/gpu_check.sh [tag]
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== smoke" | tee $OUT/summary.txt
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
echo "== pytest -m gpu (engine vs oracle, golden fixtures, reference pin, full-size and scene-size properties, data parallel)" | tee -a $OUT/summary.txt
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -8 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
echo "== bench (the driver's command, default flags)" | tee -a $OUT/summary.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err; echo "bench (driver's flags) rc=$?" | tee -a $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2>> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench.json | tee -a $OUT/summary.txt
python - <<PY | tee -a $OUT/summary.txt
import json
for f in ("bench_driver_flags.json", "bench.json"):
    d = json.load(open("$OUT/" + f)); r = d["roofline"]
    print(f, "ms_per_step", d["ms_per_step"], "| top", r["kernel"], r["us"], "us frac", r["frac"], "| boundary", r["boundary"]["ball_query_group"]["frac"], "min", r["boundary"]["ball_query_group"]["frac_min"], "| achieved_step", r["achieved_step"]["frac"], "| cpu", d["cpu_baseline"]["all_cores"]["value"], d["cpu_baseline"]["one_thread"]["value"])
PY
echo "== bench --precision bf16 (config 2's arithmetic)" | tee -a $OUT/summary.txt
timeout 900 python bench.py --precision bf16 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_bf16.json | cut -c1-300 | tee -a $OUT/summary.txt
echo "== bench, eager launches" | tee -a $OUT/summary.txt
timeout 900 python bench.py --no-graph --no-cpu-baseline --no-kernel-roofline 2>/dev/null | tee $OUT/bench_eager.json | cut -c1-260 | tee -a $OUT/summary.txt
echo "== ball query: LDS-resident kernel vs cell grid through HBM (same op, pinned path)" | tee -a $OUT/summary.txt
for p in tile cells; do CL3D_BQ_PATH=$p timeout 120 python scripts/bench_bq.py | tee -a $OUT/bench_bq.jsonl | tee -a $OUT/summary.txt; done
for p in tile cells; do CL3D_BQ_PATH=$p timeout 120 python scripts/bench_bq.py --n 1024 | tee -a $OUT/bench_bq.jsonl | tee -a $OUT/summary.txt; done
for p in tile cells; do CL3D_BQ_PATH=$p timeout 120 python scripts/bench_bq.py --mult 4.0 | tee -a $OUT/bench_bq.jsonl | tee -a $OUT/summary.txt; done
echo "== step variants: the cell-grid ball query through HBM scratch, points stored in cell order (experiment)" | tee -a $OUT/summary.txt
for v in "CL3D_BQ_PATH=cells" "CL3D_BENCH_SORTED=1"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms_per_step', d['ms_per_step'])" | tee -a $OUT/step_variants.txt | tee -a $OUT/summary.txt
done
echo "== micro-benchmarks: 2.1 M random 256-byte row gathers by row pitch; coordinates as 3 x dword vs 1 x dwordx4" | tee -a $OUT/summary.txt
for m in gather_pitch gather_xyz; do
  [ -x scripts/micro/$m ] || (cd scripts/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $m $m.hip > /dev/null 2>&1)
  timeout 120 scripts/micro/$m | tee -a $OUT/micro_gathers.txt | tee -a $OUT/summary.txt
done
echo "== micro-benchmark: issue rate of packed FP32 against scalar FMA" | tee -a $OUT/summary.txt
mkdir -p scripts/micro/var
[ -x scripts/micro/var/pk_rate ] || (cd scripts/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o var/pk_rate pk_rate.hip > /dev/null 2>&1)
timeout 120 scripts/micro/var/pk_rate | tee $OUT/micro_pk_rate.txt | tee -a $OUT/summary.txt
echo "== eager step: host enqueue time against drained time (one C-ABI call per pass)" | tee -a $OUT/summary.txt
timeout 300 python scripts/micro/eager_host.py 2>/dev/null | head -12 | tee $OUT/eager_host.txt | head -3 | tee -a $OUT/summary.txt
echo "== rocprofv3 kernel trace of the same bench command" | tee -a $OUT/summary.txt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --no-cpu-baseline --precondition 0 > $R/$OUT/rocprof.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
python scripts/kstats.py $OUT/prof/bench_kernel_stats.csv 100 40 | tee -a $OUT/summary.txt
echo "== PMC counters of the timed step (separate passes)" | tee -a $OUT/summary.txt
n=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"; do
  n=$((n+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$OUT/step_pmc$n -o pmc -- python $R/bench.py --steps 6 --warmup 2 --precondition 0 --no-cpu-baseline --no-kernel-roofline > $R/$OUT/step_pmc$n.log 2>&1)
done
python scripts/step_counters.py $OUT/step_pmc1 $OUT/step_pmc2 $OUT/step_pmc3 $OUT/step_pmc4 > $OUT/step_counters.json 2>> $OUT/summary.txt
head -c 1200 $OUT/step_counters.json | tee -a $OUT/summary.txt
echo "== L2 requests of the gather passes with the points stored in cell order (experiment)" | tee -a $OUT/summary.txt
(cd /tmp && CL3D_BENCH_SORTED=1 timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/$OUT/sorted_pmc -o pmc -- python $R/bench.py --steps 6 --warmup 2 --precondition 0 --no-cpu-baseline --no-kernel-roofline > /dev/null 2>&1)
python - <<PY | tee $OUT/sorted_points_experiment.txt | tee -a $OUT/summary.txt
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/sorted_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "pwmlp_query" in k or "pwmlp_support" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
base = json.load(open("$OUT/step_counters.json"))["kernels"]
print("points stored in cell order (CL3D_BENCH_SORTED=1) vs the bench's random order: L2 requests per launch")
for k, cs in acc.items():
    hit, miss = (sum(cs[c]) / len(cs[c]) for c in ("TCC_HIT_sum", "TCC_MISS_sum"))
    b = [v for n, v in base.items() if n.replace("void ", "").split("<")[0] == k.replace("void ", "").split("<")[0]]
    was = (b[0]["TCC_HIT_sum"] + b[0]["TCC_MISS_sum"]) / 1e6 if b else float("nan")
    print("  %-40s %.2f M (random order %.2f M), hit rate %.3f" % (k.replace("void cl3d::", "")[:40], (hit + miss) / 1e6, was, hit / (hit + miss)))
PY
echo "== PMC traffic of the ball_query+group kernels (separate passes)" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_fetch -o pmc -- python $R/scripts/pmc_kernels.py > $R/$OUT/pmc_fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_write -o pmc -- python $R/scripts/pmc_kernels.py > $R/$OUT/pmc_write.log 2>&1)
python scripts/pmc_kernels.py --parse $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json 2>> $OUT/summary.txt
head -c 1200 $OUT/pmc_traffic.json | tee -a $OUT/summary.txt
echo "== contraction A/B (engine MFMA f32 / bf16 vs the vendor library)" | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_point_gemm.py --sweep 2>/dev/null | tee $OUT/point_gemm.jsonl | cut -c1-700 | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_point_gemm.py --convs --reps 20 2>/dev/null > $OUT/convs.jsonl
echo "== other operators (bench.py --operator)" | tee -a $OUT/summary.txt
for op in pospool adaptive_weight pseudo_grid; do
  timeout 600 python bench.py --operator $op --no-cpu-baseline --no-kernel-roofline 2>/dev/null | tee $OUT/bench_$op.json | cut -c1-330 | tee -a $OUT/summary.txt
done
echo "== backbone steps (scripts/bench_backbone.py)" | tee -a $OUT/summary.txt
for c in modelnet_small modelnet_pointwisemlp s3dis_pseudogrid partnet_adaptive s3dis_pospool_deep; do
  timeout 600 python scripts/bench_backbone.py --config $c 2>/dev/null | tail -1 | tee -a $OUT/summary.txt
done
timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | tail -1 | tee -a $OUT/summary.txt
echo "== segmentation configs with the scene-segmentation head in the step (decoder without / with the concatenated tensor)" | tee -a $OUT/summary.txt
for c in s3dis_pseudogrid partnet_adaptive s3dis_pospool_deep; do
  timeout 600 python scripts/bench_backbone.py --config $c --head 2>/dev/null | tail -1 | tee -a $OUT/summary.txt
  timeout 600 python scripts/bench_backbone.py --config $c --head --decode cat 2>/dev/null | tail -1 | sed 's/^/concatenating decoder: /' | tee -a $OUT/summary.txt
done
timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --block modules 2>/dev/null | tail -1 | sed 's/^/round-1 block path (library conv + BatchNorm modules): /' | tee -a $OUT/summary.txt
echo "== steady-state kernel table of the config-2 backbone step (bf16)" | tee -a $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_bb -o bb -- python $R/scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 40 > $R/$OUT/rocprof_bb.log 2>&1)
python scripts/kstats.py $OUT/prof_bb/bb_kernel_stats.csv 47 50 | tee $OUT/backbone_steady_state.txt | head -30 | tee -a $OUT/summary.txt
echo "== data parallel on one device (2 ranks over gloo): backbone, flat exchange and two-graph overlapped exchange" | tee -a $OUT/summary.txt
# (a bare `--gpus 2`: the scripts re-launch themselves as two ranks; one device visible -> both ranks on it over gloo)
timeout 600 python scripts/bench_backbone.py --gpus 2 --config partnet_adaptive 2>/dev/null | grep '^{' | tail -1 | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_backbone.py --gpus 2 --config partnet_adaptive --overlap 2>/dev/null | grep '^{' | tail -1 | tee -a $OUT/summary.txt
timeout 600 python bench.py --gpus 2 --no-cpu-baseline --no-kernel-roofline 2>/dev/null | grep '^{' | tail -1 | cut -c1-400 | tee -a $OUT/summary.txt
echo "== two-graph step, 200 replays without the update: distinct bit patterns of the exchanged gradients (DESIGN 6)" | tee -a $OUT/summary.txt
for cfg in "" "--overlap --overlap-forks none" "--overlap --overlap-forks a" "--overlap --overlap-forks b" "--overlap --overlap-forks b --fork-mode probe"; do
  timeout 300 python scripts/bench_backbone.py --gpus 2 --config modelnet_small --warmup 1 --head $cfg --repeat-check 200 2>/dev/null | grep repeat_check | python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['varying_parameters']
print('[$cfg]', 'late', d['distinct_late'][:4], 'early', d['distinct_early'][:4], 'varying parameters', len(v), 'probe', d['probe'])" | tee -a $OUT/two_graph_repeat_check.txt | tee -a $OUT/summary.txt
done
echo "== config 2 layer by layer (--layerwise): the activated tensors between a bottleneck's layers materialised" | tee -a $OUT/summary.txt
for prec in f32 bf16; do timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision $prec --layerwise 2>/dev/null | tail -1 | tee -a $OUT/summary.txt; done
echo "== dataset-side grid subsampling, voting, sphere crops" | tee -a $OUT/summary.txt
timeout 600 python scripts/bench_dataset_grid.py 2>/dev/null | tee $OUT/bench_dataset_grid.json | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_voting.py 2>/dev/null | tee $OUT/bench_voting.json | tee -a $OUT/summary.txt
timeout 300 python scripts/bench_sphere_crop.py 2>/dev/null | tee $OUT/bench_sphere_crop.json | tee -a $OUT/summary.txt
echo "== timeline of one replay of the timed step (critical path, idle time between kernels)" | tee -a $OUT/summary.txt
(cd /tmp && rm -rf /tmp/tl && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $R/bench.py --steps 12 --warmup 4 --precondition 0 --no-cpu-baseline --no-kernel-roofline --no-step-table > /dev/null 2>&1)
python scripts/step_timeline.py "/tmp/tl/**/tl_kernel_trace.csv" | tee $OUT/step_timeline.txt | tail -3 | tee -a $OUT/summary.txt
echo "== PseudoGrid operator: per-kernel averages" | tee -a $OUT/summary.txt
(cd /tmp && rm -rf /tmp/pgp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pgp -o pg -- python $R/bench.py --operator pseudo_grid --steps 40 --warmup 5 --precondition 0 --no-cpu-baseline --no-kernel-roofline --no-step-table > /dev/null 2>&1)
cp $(find /tmp/pgp -name "pg_kernel_stats.csv" | head -1) $OUT/bench_pseudo_grid_kernel_stats.csv 2>/dev/null
python scripts/kstats.py $OUT/bench_pseudo_grid_kernel_stats.csv 45 12 | tee -a $OUT/summary.txt
# keep the merged output small: drop raw traces, keep stats and counter tables
find $OUT -type f -name "*kernel_trace*" -delete 2>/dev/null
find $OUT -type f -size +3M -delete 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt

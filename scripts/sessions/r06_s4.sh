#!/bin/bash
# Round 6 session 4: roofline tables of PosPool / AdaptiveWeight / PseudoGrid (PMC passes first, then the bench line that reads
# them), rocprofv3 --stats of each; in-launch slice sums A/B on the config-2 backbone; the whole -m gpu suite
cd "$(dirname "$0")/../.." || exit 1
TAG=r06_s4
OUT=gpurun_out/$TAG
mkdir -p $OUT profiles/r06
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
echo "== operators: PMC passes, then bench.py with its step table" | tee $OUT/summary.txt
for op in pospool adaptive_weight pseudo_grid; do
  bash scripts/pmc_bench.sh $TAG/pmc_$op $op "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" > $OUT/pmc_$op.txt 2>&1
  python scripts/step_counters.py $OUT/pmc_$op/pmc1 $OUT/pmc_$op/pmc2 $OUT/pmc_$op/pmc3 $OUT/pmc_$op/pmc4 > profiles/r06/step_counters_$op.json 2>> $OUT/summary.txt
  cp profiles/r06/step_counters_$op.json $OUT/
  timeout 900 python bench.py --operator $op --no-cpu-baseline --backbone off 2>/dev/null > $OUT/bench_$op.json
  python - <<PY | tee -a $OUT/summary.txt
import json
d = json.load(open("$OUT/bench_$op.json"))
print("$op", d["ms_per_step"], "ms", d["value"], "points/s achieved_step", d["roofline"]["achieved_step"]["frac"])
for r in d["roofline"]["step"]["kernels"]:
    print("   %-34s %2d x %7.1f us  alg %8.1f MB  hbm_frac %s  pmc %s MB  l2 %s MB" % (r["entry"], r["calls"], r["us"], r["algorithmic_bytes"]/1e6, r["hbm_frac"], round(r.get("hbm_bytes_pmc",0)/1e6,1), round(r.get("l2_bytes_pmc",0)/1e6,1)))
PY
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$op -o bench -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-roofline --backbone off --operator $op > $R/$OUT/rocprof_$op.log 2>&1)
  python scripts/kstats.py $(find $OUT/prof_$op -name "bench_kernel_stats.csv" | head -1) 345 16 | tee $OUT/kstats_$op.txt | head -14 | tee -a $OUT/summary.txt
  cp $(find $OUT/prof_$op -name "bench_kernel_stats.csv" | head -1) $OUT/bench_${op}_kernel_stats.csv
done
echo "== config-2 backbone: in-launch slice sum (64 x 64 tiles, <= 3 slices) on / off, variant library, alternating" | tee -a $OUT/summary.txt
for i in 1 2 3; do for f in 1 0; do
  CL3D_LIB=$R/scripts/micro/var/libcl3d_gemm_plan_env.so CL3D_GEMM_FUSED_SUM=$f timeout 600 python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('in-launch sum $f', d['ms_per_step'])" | tee -a $OUT/summary.txt
done; done
echo "== pytest -m gpu (whole suite)" | tee -a $OUT/summary.txt
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -6 $OUT/pytest_gpu.log | cut -c1-300 | tee -a $OUT/summary.txt
find $OUT -name "*kernel_trace*" -delete 2>/dev/null; find $OUT -type f -size +3M -delete 2>/dev/null
echo "== done" | tee -a $OUT/summary.txt

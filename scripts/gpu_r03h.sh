#!/bin/bash
# Round-3 session H: DP overlap test, points-in-cell-order experiment (step table + TCC requests)
TAG=${1:-r03h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== dp tests" | tee $OUT/summary.txt
timeout 1200 python -m pytest tests/test_dp_gpu.py -m gpu -q -x --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -6 $OUT/pytest.log | tee -a $OUT/summary.txt
for v in "" "CL3D_BENCH_SORTED=1"; do
  echo "-- $v" | tee -a $OUT/summary.txt
  env $v timeout 300 python bench.py --no-cpu-baseline --bursts 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms_per_step', d['ms_per_step'], ' '.join('%s=%.1f' % (k['entry'].replace('cl3d_',''), k['us']) for k in r['step']['kernels'][:9]))
" | tee -a $OUT/summary.txt
  (cd /tmp && env $v timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/$OUT/pmc_$(echo $v | tr -d '=') -o pmc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-roofline > /dev/null 2>&1)
done
python - <<'PY' | tee -a gpurun_out/r03h/summary.txt
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r03h/pmc_*")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if "pwmlp_query" in k or "pwmlp_support" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        hit, miss = (sum(cs[c]) / len(cs[c]) for c in ("TCC_HIT_sum", "TCC_MISS_sum"))
        print(d.split("/")[-1], k[:50], "L2 requests %.2f M, hit rate %.3f" % ((hit + miss) / 1e6, hit / (hit + miss)))
PY
find $OUT -name "*kernel_trace*" -delete; find $OUT -type f -size +2M -delete
echo "== done" | tee -a $OUT/summary.txt

#!/bin/bash
# Round 6 session 29: config 2 (bf16) under the profiler -- how much of a replayed step has NO kernel running, how much has
# exactly one, and the longest gaps with the kernels either side (what the cross-queue joins of the captured step cost)
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r06_s29
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/prof_bb -o bb -- python $R/scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 150 > $R/$OUT/rocprof_bb.log 2>&1)
grep '^{' $OUT/rocprof_bb.log | tail -1 | cut -c1-300 | tee $OUT/summary.txt
T=$(find $OUT/prof_bb -name "bb_kernel_trace.csv" | head -1)
python - "$T" <<'PY' | tee -a $OUT/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
t_end = max(e[1] for e in ev)
win = [e for e in ev if e[0] >= t_end - 300e6]
steps = sum(1 for e in win if "pwmlp_hit_coeffs_kernel" in e[2]) // 4
t0, t1 = win[0][0], max(e[1] for e in win)
# sweep: time with 0, 1, >= 2 kernels running
pts = []
for s, e, _, _ in win:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth, last, hist = 0, t0, collections.Counter()
for t, d in pts:
    hist[min(depth, 2)] += t - last
    last = t
    depth += d
tot = t1 - t0
print("window %.1f ms, %d steps, %.1f us per step" % (tot / 1e6, steps, tot / 1e3 / steps))
for k in (0, 1, 2):
    print("  %s kernels running: %.1f us per step (%.1f %%)" % ("no" if k == 0 else ("one" if k == 1 else ">= 2"), hist[k] / 1e3 / steps, 100.0 * hist[k] / tot))
# gaps: intervals with no kernel running
win.sort()
gaps = []
cur_end, cur_name = win[0][1], win[0][2]
for s, e, n, q in win[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, cur_name, n))
    if e > cur_end:
        cur_end, cur_name = e, n
import statistics
print("  gaps with no kernel running: %d per step, median %.1f us, mean %.1f us" % (len(gaps) // steps, statistics.median(g[0] for g in gaps) / 1e3, sum(g[0] for g in gaps) / len(gaps) / 1e3))
big = collections.Counter()
for g, a, b in gaps:
    if g > 4000:
        big[(a.split("(")[0][-50:], b.split("(")[0][-50:])] += g
print("  gaps > 4 us by (kernel before -> kernel after), us per step:")
for (a, b), g in big.most_common(25):
    print("    %7.1f  %s -> %s" % (g / 1e3 / steps, a, b))
queues = collections.Counter()
for s, e, n, q in win:
    queues[q] += e - s
print("  busy per queue id, us per step:", {q: round(v / 1e3 / steps, 1) for q, v in queues.items()})
PY
rm -rf $OUT/prof_bb
echo "== done" | tee -a $OUT/summary.txt

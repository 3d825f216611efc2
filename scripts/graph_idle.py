#!/usr/bin/env python3
"""How much of a replayed step has NO kernel running, exactly one, two or more -- from a rocprofv3 --kernel-trace CSV of a bench
that replays a captured step (scripts/bench_backbone.py, bench.py).  The last `--window-ms` of the trace is taken as steady state;
steps are counted by a kernel that runs a known number of times per step (`--marker`, `--per-step`).

    rocprofv3 --kernel-trace --output-format csv -d out -o bb -- python scripts/bench_backbone.py --config modelnet_pointwisemlp --precision bf16 --steps 150
    python scripts/graph_idle.py out/*/bb_kernel_trace.csv --marker pwmlp_hit_coeffs_kernel --per-step 4
"""
import argparse
import collections
import csv
import statistics


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--window-ms", type=float, default=300.0)
    ap.add_argument("--marker", default="pwmlp_hit_coeffs_kernel")
    ap.add_argument("--per-step", type=int, default=4, help="launches of the marker kernel per step (config 2: 4)")
    ap.add_argument("--gap-us", type=float, default=4.0)
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.trace)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows)
    t_end = max(e[1] for e in ev)
    win = [e for e in ev if e[0] >= t_end - args.window_ms * 1e6]
    steps = max(1, sum(1 for e in win if args.marker in e[2]) // args.per_step)
    t0, t1 = win[0][0], max(e[1] for e in win)
    pts = []
    for s, e, _, _ in win:
        pts.append((s, 1))
        pts.append((e, -1))
    pts.sort()
    depth, last, hist = 0, t0, collections.Counter()
    for t, d in pts:
        hist[min(depth, 2)] += t - last
        last = t
        depth += d
    tot = t1 - t0
    print("window %.1f ms, %d steps, %.1f us per step, %.0f launches per step" % (tot / 1e6, steps, tot / 1e3 / steps, len(win) / steps))
    for k, name in ((0, "no kernel"), (1, "one kernel"), (2, ">= 2 kernels")):
        print("  %-13s running: %7.1f us per step (%.1f %%)" % (name, hist[k] / 1e3 / steps, 100.0 * hist[k] / tot))
    win.sort()
    gaps = []
    cur_end, cur_name = win[0][1], win[0][2]
    for s, e, n, q in win[1:]:
        if s > cur_end:
            gaps.append((s - cur_end, cur_name, n))
        if e > cur_end:
            cur_end, cur_name = e, n
    print("  gaps with no kernel running: %d per step, median %.1f us, mean %.1f us"
          % (len(gaps) // steps, statistics.median(g[0] for g in gaps) / 1e3, sum(g[0] for g in gaps) / len(gaps) / 1e3))
    big = collections.Counter()
    for g, a, b in gaps:
        if g > args.gap_us * 1e3:
            big[(a.split("(")[0][-50:], b.split("(")[0][-50:])] += g
    print("  gaps > %.0f us by (kernel before -> kernel after), us per step:" % args.gap_us)
    for (a, b), g in big.most_common(20):
        print("    %7.1f  %s -> %s" % (g / 1e3 / steps, a, b))
    queues = collections.Counter()
    for s, e, n, q in win:
        queues[q] += e - s
    print("  busy per queue id, us per step:", {q: round(v / 1e3 / steps, 1) for q, v in sorted(queues.items())})


if __name__ == "__main__":
    main()

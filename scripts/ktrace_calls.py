"""Per-launch view of a rocprofv3 kernel trace (tells which layer of a backbone a slow launch belongs to).
    python scripts/ktrace_calls.py <kernel_trace.csv> <pattern> [n]            last n launches matching, in order
    python scripts/ktrace_calls.py <kernel_trace.csv> <pattern> --by-grid <window_ms> <steps>
                                                                              (kernel, grid) groups, us per step"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def grid(r):
    return (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]))


def dur(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


if len(sys.argv) > 3 and sys.argv[3] == "--by-grid":
    window, steps = float(sys.argv[4]) * 1e6, int(sys.argv[5])
    t_end = max(int(r["End_Timestamp"]) for r in rows)
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if int(r["Start_Timestamp"]) >= t_end - window:
            a = acc[(r["Kernel_Name"][:60], grid(r))]
            a[0] += 1
            a[1] += dur(r)
    for (name, g), (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{n / steps:6.1f}/step {us / n:9.1f} us {us / steps:9.1f} us/step  grid {g[0]:6d} x {g[1]:4d}  {name}")
else:
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    for r in rows[-n:]:
        g = grid(r)
        print(f"{dur(r):9.1f} us  grid {g[0]:6d} x {g[1]:4d}  lds {r.get('LDS_Block_Size', '?'):>6}  {r['Kernel_Name'][:70]}")

"""Aggregate a rocprofv3 kernel trace over its last `window_ms` milliseconds (steady state only).
    python scripts/ktrace_tail.py <kernel_trace.csv> <window_ms> <steps_in_window> [top]"""
import csv
import collections
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
window = float(sys.argv[2]) * 1e6
steps = int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 30
t_end = max(int(r["End_Timestamp"]) for r in rows)
acc = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t_end - window:
        continue
    a = acc[r["Kernel_Name"]]
    a[0] += 1
    a[1] += (e - s) / 1e3
    tot += (e - s) / 1e3
print(f"window {window / 1e6:.1f} ms: GPU busy {tot / 1e3:.2f} ms = {tot / steps:.1f} us/step over {steps} steps")
for name, (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{n / steps:8.1f}/step {us / n:9.1f} us {us / steps:10.1f} us/step {100 * us / tot:6.2f}%  {name[:90]}")

"""Print a rocprofv3 kernel_stats.csv compactly: calls, avg us, total ms, short name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU time {tot/1e6:.3f} ms  ({tot/1e3/steps:.1f} us per step over {steps:g} steps)")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    name = r["Name"].replace("void ", "")
    name = name[:70]
    print(f'{int(r["Calls"]):6d} {float(r["AverageNs"])/1e3:10.1f} us {float(r["TotalDurationNs"])/1e3/steps:10.1f} us/step  {float(r["Percentage"]):6.2f}%  {name}')

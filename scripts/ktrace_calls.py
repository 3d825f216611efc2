"""List the last `n` launches of every kernel whose name contains `pattern`, in launch order, with grid size and
duration (tells which layer of a backbone a slow launch belongs to).
    python scripts/ktrace_calls.py <kernel_trace.csv> <pattern> [n]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-n:]:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{us:9.1f} us  grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):6d} x {r['Grid_Size_Y']:>4}  "
          f"lds {r.get('LDS_Block_Size', '?'):>6}  {r['Kernel_Name'][:70]}")

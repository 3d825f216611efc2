"""Re-attach the PMC columns of a saved bench line (roofline.step.kernels[*].hbm_bytes_pmc / l2_bytes_pmc) from a
step_counters.json of the SAME session.  bench.py reads the newest committed profiles/rNN/step_counters.json while it
runs, i.e. the previous session's counters; after a session's files are copied to profiles/rNN/ this puts the
session's own counters next to its own timings.

    python scripts/refresh_step_pmc.py profiles/r02/bench_pointwisemlp.json profiles/r02/step_counters.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import L2_PEAK, entry_kernels  # noqa: E402


def main():
    line_path, counters_path = sys.argv[1], sys.argv[2]
    line = json.load(open(line_path))
    counters = json.load(open(counters_path))["kernels"]
    step = line["roofline"]["step"]
    for row in step["kernels"]:
        hb = lb = 0.0
        found = False
        for kname, rec in counters.items():
            if any(kname.replace("cl3d::", "").startswith(pref) for pref in entry_kernels(row["entry"], counters)):
                hb += rec.get("hbm_bytes", 0.0)
                lb += rec.get("l2_bytes", 0.0)
                found = True
        row["kernels"] = entry_kernels(row["entry"], counters)
        row.pop("hbm_bytes_pmc", None)
        row.pop("l2_bytes_pmc", None)
        row.pop("l2_frac", None)
        if found:
            row["hbm_bytes_pmc"], row["l2_bytes_pmc"] = int(hb), int(lb)
            row["l2_frac"] = round(lb / (row["us"] * 1e-6) / L2_PEAK, 4) if row["us"] > 0 else None
    step["pmc_source"] = os.path.relpath(counters_path)
    top = line["roofline"]
    for row in step["kernels"]:  # the top-level block repeats one row's counters
        if row["entry"] == top.get("entry"):
            top["traffic"], top["l2_frac"] = row.get("hbm_bytes_pmc"), row.get("l2_frac")
    json.dump(line, open(line_path, "w"))
    print("refreshed", line_path, "from", counters_path)


if __name__ == "__main__":
    main()

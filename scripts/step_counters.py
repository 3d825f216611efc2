"""PMC counters of the bench step, per kernel -> profiles/rNN/step_counters.json (read back by bench.py's
`roofline.step`).

Collection (separate passes, kernel-trace only, as MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes):
    bash scripts/pmc_bench.sh <tag> pointwisemlp "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
        "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"
    python scripts/step_counters.py gpurun_out/<tag>/pmc1 gpurun_out/<tag>/pmc2 gpurun_out/<tag>/pmc3 [...] > profiles/rNN/step_counters.json

Units and corrections: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced
read, so hbm_bytes = 2 * fetch + write (the guide's correction; WRITE_SIZE is taken raw); l2_bytes = (TCC_HIT_sum +
TCC_MISS_sum) * 128 B (requests of one 128-byte line each).  Values are averages per launch.
"""
import collections
import csv
import glob
import json
import os
import sys


def collect(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if not name.startswith("cl3d::"):
                    continue
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    acc = collect(sys.argv[1:])
    out = {"note": __doc__.split("Units and corrections:")[1].strip(), "kernels": {}}
    for k in sorted(acc):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        rec = {"launches_sampled": max(len(v) for v in acc[k].values())}
        rec.update({n: round(v, 1) for n, v in c.items()})
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            rec["hbm_bytes"] = 2 * c.get("FETCH_SIZE", 0.0) * 1024 + c.get("WRITE_SIZE", 0.0) * 1024
        if "TCC_HIT_sum" in c or "TCC_MISS_sum" in c:
            req = c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0)
            rec["l2_bytes"] = req * 128
            rec["l2_hit_rate"] = round(c.get("TCC_HIT_sum", 0.0) / req, 4) if req else None
        if c.get("SQ_WAVE_CYCLES"):
            rec["wait_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
        out["kernels"][k] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

"""Launch each kernel of the ball_query+group boundary a few times at the metric shape (for rocprofv3 --pmc).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -o pmc -- python scripts/pmc_kernels.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/write -o pmc -- python scripts/pmc_kernels.py
    python scripts/pmc_kernels.py --parse out/fetch out/write > profiles/rNN/pmc_traffic.json
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np
    import torch
    from bench import synth_batch
    from closerlook3d_amd import _ext
    B, N, K, C = 16, 4096, 32, 64
    radius = float((1.5 * K * 3 / (4 * np.pi * N)) ** (1 / 3))
    xyz, mask, feats = (torch.from_numpy(a).cuda() for a in synth_batch(B, N, C, 1000))
    idx, _ = _ext.masked_ordered_ball_query(xyz, xyz, mask, mask, radius, K)
    g = torch.randn(B, C, N, K, device="cuda")
    for _ in range(3):
        _ext.masked_ordered_ball_query(xyz, xyz, mask, mask, radius, K)
        out = _ext.group_points(feats, idx)
        del out
        _ext.group_points_grad(g, idx, N)
        _ext.group_xyz_features(xyz, xyz, None, idx, radius, True)
    torch.cuda.synchronize()


def parse(fetch_dir, write_dir):
    def collect(d, counter):
        acc = {}
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if r.get("Counter_Name") != counter:
                    continue
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                acc.setdefault(name, []).append(float(r["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in acc.items()}
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc, separate passes; FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half "
                   "the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM), so fetch_bytes_corrected = 2 * raw; "
                   "WRITE_SIZE is uncalibrated there and is reported raw. Per launch, metric shape B=16,N=4096,K=32,C=64.",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("cl3d::"):
            continue
        f, w = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        out["kernels"][k] = {"fetch_bytes_raw": f, "fetch_bytes_corrected": 2 * f, "write_bytes_raw": w,
                             "hbm_bytes": 2 * f + w}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--parse":
        parse(sys.argv[2], sys.argv[3])
    else:
        run()

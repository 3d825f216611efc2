"""Time masked_ordered_ball_query at the metric shape (or --n/--m/--k) under the current CL3D_BQ_PATH; one JSON line.
--tuned: the path the engine's callers take (closerlook3d_amd.pt_utils: both applicable paths timed once per (sizes, radius)
key, the faster one kept) -- the line then names the path that was kept and the two timings of the tuning pass."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from closerlook3d_amd import _ext  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=16)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--m", type=int, default=0)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--mult", type=float, default=1.5)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--tuned", action="store_true", help="through pt_utils' measured dispatch instead of the library's choice by size")
    ap.add_argument("--dump", default="", help="write idx of the first call to this .npy (bit-exactness of kernel variants)")
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    s = torch.from_numpy(rng.random((a.b, a.n, 3), dtype=np.float32)).cuda()
    sm = torch.ones((a.b, a.n), dtype=torch.int32, device="cuda")
    if a.m:
        q = s[:, :a.m].contiguous() + 0.001
        qm = torch.ones((a.b, a.m), dtype=torch.int32, device="cuda")
    else:
        q, qm = s, sm
    r = float((a.mult * a.k * 3 / (4 * np.pi * a.n)) ** (1 / 3))
    path, tuning = 0, None
    if a.tuned:
        from closerlook3d_amd import pt_utils
        path = pt_utils._bq_path(q, s, qm, sm, r, a.k)
        tuning = {k: round(v, 2) for k, v in pt_utils._BQ_PATH_TABLE[pt_utils.bq_tune_key(0, a.b, a.m or a.n, a.n, a.k, r)][1].items()}
    for _ in range(5):
        out = _ext.masked_ordered_ball_query(q, s, qm, sm, r, a.k, path)
    torch.cuda.synchronize()
    if a.dump:
        np.save(a.dump, torch.stack(out).cpu().numpy())
    ts = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _ext.masked_ordered_ball_query(q, s, qm, sm, r, a.k, path)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(json.dumps({"op": "masked_ordered_ball_query", "path": os.environ.get("CL3D_BQ_PATH", "auto") if not a.tuned else {0: "library", 1: "tile", 2: "cells"}[path],
                      "tuning_us": tuning, "B": a.b, "N": a.n,
                      "M": a.m or a.n, "K": a.k, "mult": a.mult, "us_median": round(ts[len(ts) // 2], 2),
                      "us_min": round(ts[0], 2), "us_max": round(ts[-1], 2)}))


if __name__ == "__main__":
    main()

"""Dataset-side grid subsampling: points/s of the engine (cloud resident in HBM) next to the reference's own C++
(oracle/_ref/libgrid_dataset_ref.so, one host thread -- the reference runs it per sample in its loader workers).
    python scripts/bench_dataset_grid.py [--points 2000000] [--dl 0.04]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=2_000_000)
    ap.add_argument("--dl", type=float, default=0.04)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    from closerlook3d_amd.data_utils import grid_subsampling
    rng = np.random.default_rng(0)
    n = args.points
    p = (rng.random((n, 3), dtype=np.float32) * np.float32([8, 6, 3])).astype(np.float32)
    f = rng.random((n, 4), dtype=np.float32)
    l = rng.integers(0, 13, (n, 1)).astype(np.int32)
    tp, tf, tl = (torch.from_numpy(a).cuda() for a in (p, f, l))
    for _ in range(2):
        out = grid_subsampling(tp, tf, tl, sampleDl=args.dl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        out = grid_subsampling(tp, tf, tl, sampleDl=args.dl)
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t0) / args.iters
    line = {"op": "dataset grid_subsampling (xyz + 4 features + 1 label column)", "points": n, "sampleDl": args.dl,
            "voxels": int(out[0].shape[0]), "gpu_ms": round(gpu * 1e3, 3), "gpu_points_per_s": round(n / gpu, 1)}
    from bench import cpu_baseline_dataset_grid   # the reference build is timed by bench.py's cpu_baseline leg only
    ref = cpu_baseline_dataset_grid(p, f, l, args.dl)
    if ref is None:
        line["reference_cpu_ms"] = None
    else:
        cpu, m = ref
        line.update(reference_cpu_ms=round(cpu * 1e3, 1), reference_cpu_points_per_s=round(n / cpu, 1),
                    reference_voxels=m, speedup=round(cpu / gpu, 1))
    print(json.dumps(line))


if __name__ == "__main__":
    main()

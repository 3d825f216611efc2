"""Time masked_grid_subsampling (one workgroup per cloud) at the config-2 pyramid's sizes; one JSON line per size.
CL3D_LIB selects a variant build (scripts/micro/kernel_variants.py: sub_bitonic = the round-5 bitonic network)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from closerlook3d_amd import _ext  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    first = None
    for N, m, dl in ((4096, 1024, 0.04), (1024, 256, 0.08), (256, 64, 0.16), (16384, 4096, 0.02)):
        B = 16
        xyz = torch.from_numpy(rng.random((B, N, 3), dtype=np.float32)).cuda()
        mask = torch.ones((B, N), dtype=torch.int32, device="cuda")
        mask[:, int(0.9 * N):] = 0
        for _ in range(5):
            out = _ext.masked_grid_subsampling(xyz, mask, m, dl)
        torch.cuda.synchronize()
        ts = []
        for _ in range(50):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _ext.masked_grid_subsampling(xyz, mask, m, dl)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        bits = int(out[0].view(torch.int32).long().sum()) ^ int(out[1].long().sum())
        print(json.dumps({"op": "masked_grid_subsampling", "lib": os.path.basename(os.environ.get("CL3D_LIB", "libcl3d.so")), "B": B,
                          "N": N, "m": m, "dl": dl, "us_median": round(ts[len(ts) // 2], 2), "us_min": round(ts[0], 2),
                          "checksum": bits}))


if __name__ == "__main__":
    main()

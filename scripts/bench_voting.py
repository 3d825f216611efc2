"""Vote bookkeeping: the device accumulator against the reference's host loop (oracle/voting.py restates it;
the reference's is inline in train_s3dis_dist.py:357-369).  S3DIS-like sizes: 13 classes, batches of 8 crops of
15 000 points, two scenes of 600 000 and 350 000 sub-sampled points.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from closerlook3d_amd import voting  # noqa: E402
from bench import cpu_baseline_voting  # noqa: E402  (the oracle is timed by bench.py's cpu_baseline leg only)


def main():
    rng = np.random.default_rng(0)
    C, B, N, sizes = 13, 8, 15000, [600000, 350000]
    batches = []
    for _ in range(6):
        label = rng.integers(0, 2, size=B)
        inds = np.stack([rng.permutation(sizes[c])[:N] for c in label]).astype(np.int64)
        batches.append((rng.normal(size=(B, C, N)).astype(np.float32), np.ones((B, N), np.int32), inds, label))
    dev = torch.device("cuda")
    votes = voting.VoteAccumulator(C, sizes, device=dev)
    on_dev = [(torch.from_numpy(p).to(dev), torch.from_numpy(m).to(dev), torch.from_numpy(i).to(dev), l.tolist())
              for p, m, i, l in batches]
    votes.update(*on_dev[0])  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in on_dev:
        votes.update(*b)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / len(on_dev) * 1e3
    # the reference first copies every element's tensors to the host (not counted here), then runs its loop
    cpu_ms = cpu_baseline_voting(batches[:2], C, sizes) * 1e3
    print(json.dumps({"op": "vote update of one batch (8 crops x 15000 points, 13 classes; scenes of 600k / 350k points)",
                      "gpu_ms": round(gpu_ms, 3), "reference_host_loop_ms": round(cpu_ms, 1),
                      "ratio": round(cpu_ms / gpu_ms, 1)}))


if __name__ == "__main__":
    main()

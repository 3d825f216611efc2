"""Sphere crop of an S3DIS-sized scene: the device pass against scikit-learn's KD-tree on one host core (what a
DataLoader worker of the reference runs per sample, datasets/S3DIS.py:296-306).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from closerlook3d_amd.sphere_crop import SceneCropper  # noqa: E402


def main():
    from sklearn.neighbors import KDTree
    rng = np.random.default_rng(0)
    n, radius, num_points = 800000, 2.0, 15000
    pts = rng.uniform([0, 0, 0], [40, 25, 3], size=(n, 3)).astype(np.float32)   # ~4 cm grid density of a floor
    picks = pts[rng.integers(0, n, size=20)] + rng.normal(scale=0.2, size=(20, 3)).astype(np.float32)
    scene = SceneCropper(pts, in_radius=radius, num_points=num_points, device="cuda")
    got = scene.query(picks[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in picks:
        s = scene.crop(p)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / len(picks) * 1e3
    batch = picks[:8]
    scene.crop_batch(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        sb = scene.crop_batch(batch)
    torch.cuda.synchronize()
    batch_ms = (time.perf_counter() - t0) / 5 / len(batch) * 1e3
    tree = KDTree(pts, leaf_size=50)
    t0 = time.perf_counter()
    for p in picks:
        want = tree.query_radius(p.reshape(1, -1), r=radius, return_distance=True, sort_results=True)[0][0][:num_points]
    cpu_ms = (time.perf_counter() - t0) / len(picks) * 1e3
    same = bool(np.array_equal(scene.query(picks[-1]).cpu().numpy(), want))
    print(json.dumps({"op": f"sphere crop: radius {radius} m in a scene of {n} points, keep the {num_points} nearest "
                            f"({int(got.numel())} kept at the first pick)", "gpu_ms_per_sample": round(gpu_ms, 3), "gpu_ms_per_sample_batch_of_8": round(batch_ms, 3),
                      "kdtree_host_ms_per_query": round(cpu_ms, 2), "kdtree_over_batched": round(cpu_ms / batch_ms, 2),
                      "identical_index_list": same}))


if __name__ == "__main__":
    main()

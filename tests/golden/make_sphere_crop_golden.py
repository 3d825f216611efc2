"""Golden radius queries for closerlook3d_amd/sphere_crop.py from scikit-learn's KDTree, called as the reference's
S3DIS dataset calls it (datasets/S3DIS.py:300-306):

    KDTree(points, leaf_size=50).query_radius(pick, r=in_radius, return_distance=True, sort_results=True)[0][0]

    python tests/golden/make_sphere_crop_golden.py     ->  tests/golden/sphere_crop.npz
"""
import os

import numpy as np
from sklearn.neighbors import KDTree

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sphere_crop.npz")


def main():
    rng = np.random.default_rng(31)
    # a "room": points on a floor, two walls and some clutter, float32 like the sub-sampled clouds
    n = 20000
    pts = rng.uniform([0, 0, 0], [12, 9, 3], size=(n, 3))
    pts[: n // 3, 2] = rng.normal(0.02, 0.01, n // 3)
    pts[n // 3: n // 2, 0] = rng.normal(0.05, 0.01, n // 2 - n // 3)
    pts = pts.astype(np.float32)
    tree = KDTree(pts, leaf_size=50)   # leaf_size as in S3DIS.py:181
    out = {"points": pts, "in_radius": 2.0}
    picks = []
    for i in range(6):
        centre = pts[rng.integers(0, n)].reshape(1, -1)
        noise = rng.normal(scale=0.2, size=centre.shape)
        pick = centre + noise.astype(centre.dtype)    # :297-298 (float32 + float32)
        picks.append(pick[0])
        out[f"inds{i}"] = tree.query_radius(pick, r=2.0, return_distance=True, sort_results=True)[0][0]
    out["picks"] = np.stack(picks)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, [len(out[f"inds{i}"]) for i in range(6)])


if __name__ == "__main__":
    main()

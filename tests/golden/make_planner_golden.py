"""Golden plan for closerlook3d_amd.sphere_crop.EpochPlanner / oracle/planner.py.

The reference's planner sits inside `S3DIS.__init__` (datasets/S3DIS.py:212-253) behind the dataset files, so it
cannot be imported and run here; this script executes THE REFERENCE'S LITERAL NUMPY EXPRESSIONS for the potential
update (:247-251) on scikit-learn KDTree queries (:239-242) under this container's NumPy (2.2: NEP 50 promotion) and
stores the resulting plan as the 'nep50' golden.  The 'legacy' golden (NumPy < 2 value-based casting: float32 array
with a float64 scalar stays float32 -- no switch for it exists in NumPy 2.2) comes from oracle/planner.py's explicit
restatement and is cross-checked here against the literal expressions with the scalar pre-cast to float32, which is
what value-based casting did.

    python tests/golden/make_planner_golden.py     ->  tests/golden/planner.npz
"""
import os
import sys

import numpy as np
from sklearn.neighbors import KDTree

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import planner as op  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "planner.npz")


def literal(sub_points, potentials, noise, in_radius, num_points, scalar):
    trees = [KDTree(p, leaf_size=50) for p in sub_points]
    potentials = [p.copy() for p in potentials]
    min_potentials = [float(np.min(p)) for p in potentials]
    cloud_inds, point_inds = [], []
    for st in range(len(noise)):
        cloud_ind = int(np.argmin(min_potentials))
        point_ind = np.argmin(potentials[cloud_ind])
        cloud_inds.append(cloud_ind)
        point_inds.append(int(point_ind))
        points = np.array(trees[cloud_ind].data, copy=False)
        center_point = points[point_ind, :].reshape(1, -1)
        pick_point = center_point + noise[st].reshape(1, -1).astype(center_point.dtype)
        query_inds = trees[cloud_ind].query_radius(pick_point, r=in_radius, return_distance=True, sort_results=True)[0][0]
        if num_points < query_inds.shape[0]:
            query_inds = query_inds[:num_points]
        dists = np.sum(np.square((points[query_inds] - pick_point).astype(np.float32)), axis=1)
        tukeys = np.square(1 - dists / scalar)
        tukeys[dists > scalar] = 0
        potentials[cloud_ind][query_inds] += tukeys
        min_potentials[cloud_ind] = float(np.min(potentials[cloud_ind]))
    return cloud_inds, point_inds, potentials


def main():
    rng = np.random.default_rng(17)
    clouds = []
    for n, ext in ((6000, (8.0, 6.0, 3.0)), (4500, (5.0, 7.0, 3.0)), (3000, (4.0, 4.0, 3.0))):
        p = rng.uniform([0, 0, 0], ext, size=(n, 3))
        p[: n // 3, 2] = rng.normal(0.02, 0.01, n // 3)
        clouds.append(p.astype(np.float32))
    in_radius, num_points, steps = 1.5, 900, 120
    potentials = [rng.random(len(c)) * 1e-3 for c in clouds]
    noise = rng.normal(scale=in_radius / 10, size=(steps, 3))
    out = {"in_radius": in_radius, "num_points": num_points, "noise": noise}
    for i, (c, p) in enumerate(zip(clouds, potentials)):
        out[f"cloud{i}"], out[f"potential{i}"] = c, p
    # NEP 50: the literal expressions as this NumPy evaluates them
    ci, pi, pot = literal(clouds, potentials, noise, in_radius, num_points, np.square(in_radius))
    oc, opi, opot = op.plan(clouds, potentials, noise, in_radius, num_points, "nep50")
    assert ci == oc and pi == opi and all(np.array_equal(a, b) for a, b in zip(pot, opot)), "oracle != literal (nep50)"
    out["nep50_cloud_inds"], out["nep50_point_inds"] = np.array(ci), np.array(pi)
    out["nep50_potential_sums"] = np.array([p.sum() for p in pot])
    # legacy: value-based casting turned the float64 scalar into float32
    ci, pi, pot = literal(clouds, potentials, noise, in_radius, num_points, np.float32(np.square(in_radius)))
    oc, opi, opot = op.plan(clouds, potentials, noise, in_radius, num_points, "legacy")
    assert ci == oc and pi == opi and all(np.array_equal(a, b) for a, b in zip(pot, opot)), "oracle != literal (legacy)"
    out["legacy_cloud_inds"], out["legacy_point_inds"] = np.array(ci), np.array(pi)
    out["legacy_potential_sums"] = np.array([p.sum() for p in pot])
    np.savez_compressed(OUT, **out)
    same = int((out["legacy_point_inds"] == out["nep50_point_inds"]).sum())
    print("wrote", OUT, "steps", steps, "identical picks under both promotion rules:", same)


if __name__ == "__main__":
    main()

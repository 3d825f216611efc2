"""TEST INFRASTRUCTURE -- numpy restatement of the reference's S3DIS epoch planner (datasets/S3DIS.py:212-253), the
checker for closerlook3d_amd.sphere_crop.EpochPlanner.  Line by line, with the dtype of every intermediate written out
for both NumPy promotion regimes ('legacy' = NumPy < 2, value-based casting, the reference's era; 'nep50' = NumPy >= 2,
where the literal expressions of the reference promote to float64).  The radius query is scikit-learn's KDTree, called
as the reference calls it (:181, :235-238), when the package is there; otherwise the same list by brute force.
"""
import numpy as np


def _radius_sorted(points64, tree, pick, r):
    if tree is not None:
        return tree.query_radius(pick.reshape(1, -1), r=r, return_distance=True, sort_results=True)[0][0]
    d = points64 - pick
    rdist = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    inside = np.nonzero(rdist <= r * r)[0]
    return inside[np.argsort(np.sqrt(rdist[inside]), kind="stable")]


def plan(sub_points, potentials, noise, in_radius, num_points, promotion="legacy", use_sklearn=True):
    """sub_points: list of [n_i, 3] float32 clouds; potentials: list of [n_i] float64 (copied); noise [steps, 3] float64.
    Returns (cloud_inds, point_inds, potentials after the last step)."""
    trees = [None] * len(sub_points)
    if use_sklearn:
        try:
            from sklearn.neighbors import KDTree
            trees = [KDTree(p, leaf_size=50) for p in sub_points]                      # :181
        except ImportError:
            pass
    pts64 = [np.asarray(p, dtype=np.float64) for p in sub_points]                       # KDTree keeps float64 copies
    potentials = [np.array(p, dtype=np.float64) for p in potentials]
    min_potentials = [float(np.min(p)) for p in potentials]                             # :222
    cloud_inds, point_inds = [], []
    r2 = float(np.square(in_radius))
    for step in range(len(noise)):
        cloud_ind = int(np.argmin(min_potentials))                                      # :228
        point_ind = int(np.argmin(potentials[cloud_ind]))                               # :229
        cloud_inds.append(cloud_ind)
        point_inds.append(point_ind)
        points = pts64[cloud_ind]                                                       # :233 (tree.data is float64)
        center_point = points[point_ind, :].reshape(1, -1)
        pick_point = center_point + noise[step].reshape(1, -1).astype(center_point.dtype)  # :235-237
        query_inds = _radius_sorted(points, trees[cloud_ind], pick_point[0], in_radius)  # :239-242
        if num_points < query_inds.shape[0]:
            query_inds = query_inds[:num_points]                                        # :243-245
        diff32 = (points[query_inds] - pick_point).astype(np.float32)                   # :247
        sq = diff32 * diff32
        dists = (sq[:, 0] + sq[:, 1]) + sq[:, 2]                                        # np.sum(axis=1) over 3 float32
        if promotion == "legacy":   # float32 array / float64 scalar -> float32 (value-based casting, NumPy < 2)
            t = np.float32(1.0) - dists / np.float32(r2)
            tukeys = t * t
            tukeys[dists > np.float32(r2)] = 0
        else:                       # NEP 50: the float64 scalar wins
            t = 1.0 - dists.astype(np.float64) / r2
            tukeys = t * t
            tukeys[dists.astype(np.float64) > r2] = 0
        potentials[cloud_ind][query_inds] += tukeys                                     # :250
        min_potentials[cloud_ind] = float(np.min(potentials[cloud_ind]))                # :251
    return cloud_inds, point_inds, potentials

"""SURVEY 8(f) rank 2, the epoch-planning half (reference datasets/S3DIS.py:212-253 potentials / Tukey update, :262-270
projection): closerlook3d_amd.sphere_crop.EpochPlanner and SceneCropper.project against (i) tests/golden/planner.npz --
plans produced by the reference's literal numpy expressions on scikit-learn KDTree queries
(tests/golden/make_planner_golden.py), under both NumPy promotion regimes -- and (ii) oracle/planner.py run live."""
import os

import numpy as np
import pytest
import torch

from closerlook3d_amd.sphere_crop import EpochPlanner, SceneCropper
from oracle import planner as op

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "planner.npz")
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def _load():
    g = np.load(GOLDEN)
    clouds = [g[f"cloud{i}"] for i in range(3)]
    pots = [g[f"potential{i}"] for i in range(3)]
    return g, clouds, pots


@pytest.mark.parametrize("promotion", ["legacy", "nep50"])
def test_oracle_reproduces_the_literal_reference_plan(promotion):
    g, clouds, pots = _load()
    ci, pi, pot = op.plan(clouds, pots, g["noise"], float(g["in_radius"]), int(g["num_points"]), promotion, use_sklearn=False)
    assert ci == g[f"{promotion}_cloud_inds"].tolist() and pi == g[f"{promotion}_point_inds"].tolist()
    assert np.array_equal(np.array([p.sum() for p in pot]), g[f"{promotion}_potential_sums"])


def test_the_two_promotion_rules_really_differ():
    g, _, _ = _load()
    assert not np.array_equal(g["legacy_potential_sums"], g["nep50_potential_sums"])


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("promotion", ["legacy", "nep50"])
def test_device_planner_matches_golden_plan(device, promotion):
    g, clouds, pots = _load()
    scenes = [SceneCropper(c, in_radius=float(g["in_radius"]), num_points=int(g["num_points"]), device=device) for c in clouds]
    planner = EpochPlanner(scenes, pots, promotion=promotion)
    ci, pi, picks = planner.plan(g["noise"])
    assert ci == g[f"{promotion}_cloud_inds"].tolist()
    assert pi == g[f"{promotion}_point_inds"].tolist()
    got = np.array([float(p.sum()) for p in planner.potentials])
    assert np.allclose(got, g[f"{promotion}_potential_sums"], rtol=1e-12, atol=0)  # same terms; torch sums pairwise
    # pick = centre + noise in float64
    want0 = clouds[ci[0]][pi[0]].astype(np.float64) + g["noise"][0]
    assert np.array_equal(picks[0].numpy(), want0)


@pytest.mark.parametrize("device", DEVICES)
def test_planner_feeds_the_batched_cropper(device):
    """The plan's pick points go straight into crop_batch: every sample's centre is within in_radius / 10 * few sigma of
    a scene point and its first index list equals the single-pick query."""
    g, clouds, pots = _load()
    scenes = [SceneCropper(c, in_radius=float(g["in_radius"]), num_points=int(g["num_points"]), device=device) for c in clouds]
    planner = EpochPlanner(scenes, pots)
    ci, pi, picks = planner.plan(g["noise"][:8])
    for scene_id in set(ci):
        rows = [k for k in range(8) if ci[k] == scene_id]
        batch = scenes[scene_id].crop_batch(picks[rows])
        assert batch["points"].shape[0] == len(rows)
        for r, k in enumerate(rows):
            single = scenes[scene_id].query(picks[k])
            kept = int(batch["mask"][r].sum())
            assert kept == min(single.numel(), int(g["num_points"]))
            assert torch.equal(torch.sort(batch["input_inds"][r][:kept]).values, torch.sort(single).values)


@pytest.mark.parametrize("device", DEVICES)
def test_projection_matches_kdtree(device):
    sklearn_neighbors = pytest.importorskip("sklearn.neighbors")
    rng = np.random.default_rng(5)
    full = rng.uniform([0, 0, 0], [6, 5, 3], size=(20000, 3)).astype(np.float32)
    sub = full[rng.choice(len(full), 3000, replace=False)]
    want = np.squeeze(sklearn_neighbors.KDTree(sub, leaf_size=50).query(full, return_distance=False)).astype(np.int32)
    got = SceneCropper(sub, device=device).project(full).cpu().numpy()
    assert got.dtype == np.int32 and np.array_equal(got, want)

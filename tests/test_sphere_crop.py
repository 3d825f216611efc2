"""Sphere crops on the device (second half): closerlook3d_amd/sphere_crop.py against
scikit-learn's KDTree called the way the reference's S3DIS dataset calls it -- committed golden index lists
(tests/golden/sphere_crop.npz) and, on the CPU, the library itself on fresh seeds."""
import os

import numpy as np
import pytest
import torch

from closerlook3d_amd.sphere_crop import SceneCropper

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "sphere_crop.npz")
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("device", DEVICES)
def test_query_matches_kdtree_golden(device):
    g = np.load(GOLDEN)
    scene = SceneCropper(g["points"], in_radius=float(g["in_radius"]), num_points=1000, device=device)
    for i, pick in enumerate(g["picks"]):
        want = g[f"inds{i}"]
        got_all = scene.query(pick, limit=False).cpu().numpy()
        assert np.array_equal(got_all, want), f"pick {i}: sorted in-radius list differs from the KD-tree's"
        assert np.array_equal(scene.query(pick).cpu().numpy(), want[:1000])


def test_query_matches_scikit_learn_directly():
    sklearn_neighbors = pytest.importorskip("sklearn.neighbors")
    rng = np.random.default_rng(8)
    pts = rng.normal(size=(30000, 3)).astype(np.float32) * np.array([3.0, 2.0, 0.7], dtype=np.float32)
    tree = sklearn_neighbors.KDTree(pts, leaf_size=50)
    scene = SceneCropper(pts, in_radius=0.8, num_points=500, device="cpu")
    for _ in range(20):
        pick = pts[rng.integers(0, len(pts))] + rng.normal(scale=0.08, size=3).astype(np.float32)
        want = tree.query_radius(pick.reshape(1, -1), r=0.8, return_distance=True, sort_results=True)[0][0]
        assert np.array_equal(scene.query(pick, limit=False).numpy(), want)
    # a pick point far outside the scene
    assert scene.query(np.array([50.0, 50.0, 50.0]), limit=False).numel() == 0


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("num_points", [700, 6000])
def test_crop_assembles_the_sample_like_the_reference(device, num_points):
    """num_points below the in-radius count: the nearest num_points, shuffled, mask all ones; above: all of them,
    shuffled, padded by re-drawn valid points under mask 0 (S3DIS.py:304-314); centred float32 coordinates,
    height, colours and labels gathered with the same indices (:316-327)."""
    g = np.load(GOLDEN)
    pts = g["points"]
    rng = np.random.default_rng(3)
    colors = rng.uniform(size=(len(pts), 3)).astype(np.float32)
    labels = rng.integers(0, 13, size=len(pts)).astype(np.int32)
    scene = SceneCropper(pts, colors, labels, in_radius=float(g["in_radius"]), num_points=num_points, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(1)
    for i in (0, 3):
        pick, want = g["picks"][i], g[f"inds{i}"]
        s = {k: v.cpu().numpy() for k, v in scene.crop(pick, generator=gen).items()}
        inds, mask = s["input_inds"], s["mask"]
        assert inds.shape == (num_points,) and mask.dtype == np.int32
        if len(want) >= num_points:
            assert mask.all() and np.array_equal(np.sort(inds), np.sort(want[:num_points]))
            assert not np.array_equal(inds, want[:num_points]), "the sample must be shuffled"
        else:
            cur = len(want)
            assert mask[:cur].all() and not mask[cur:].any()
            assert np.array_equal(np.sort(inds[:cur]), np.sort(want))
            assert np.isin(inds[cur:], want).all()
        centred = (pts[inds].astype(np.float64) - pick.astype(np.float64)).astype(np.float32)
        assert np.array_equal(s["points"].view(np.uint32), centred.view(np.uint32))
        assert np.array_equal(s["height"], pts[inds][:, 2:])
        assert np.array_equal(s["colors"], colors[inds]) and np.array_equal(s["labels"], labels[inds].astype(np.int64))


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("num_points", [700, 6000])
def test_batched_crops_equal_single_crops(device, num_points):
    g = np.load(GOLDEN)
    pts = g["points"]
    labels = (np.arange(len(pts)) % 13).astype(np.int32)
    scene = SceneCropper(pts, None, labels, in_radius=float(g["in_radius"]), num_points=num_points, device=device)
    picks = g["picks"]
    cols, rows, kept = scene.query_batch(picks)
    cols, rows, kept = cols.cpu().numpy(), rows.cpu().numpy(), kept.cpu().numpy()
    for b in range(len(picks)):
        want = g[f"inds{b}"][:num_points]
        assert kept[b] == len(want) and np.array_equal(cols[rows == b], want)
    gen = torch.Generator(device=device)
    gen.manual_seed(2)
    s = {k: v.cpu().numpy() for k, v in scene.crop_batch(picks, generator=gen).items()}
    assert s["input_inds"].shape == (len(picks), num_points)
    for b in range(len(picks)):
        want = g[f"inds{b}"][:num_points]
        cur = len(want)
        inds, mask = s["input_inds"][b], s["mask"][b]
        assert mask[:cur].all() and not mask[cur:].any()
        assert np.array_equal(np.sort(inds[:cur]), np.sort(want)) and np.isin(inds[cur:], want).all()
        if cur > 50:
            assert not np.array_equal(inds[:cur], want), "the sample must be shuffled"
        if cur < num_points:
            assert len(np.unique(inds[cur:])) > 1, "padding re-draws valid points at random"
        centred = (pts[inds].astype(np.float64) - picks[b].astype(np.float64)).astype(np.float32)
        assert np.array_equal(s["points"][b].view(np.uint32), centred.view(np.uint32))
        assert np.array_equal(s["height"][b], pts[inds][:, 2:])
        assert np.array_equal(s["labels"][b], labels[inds].astype(np.int64))


@pytest.mark.gpu
def test_native_query_with_a_sort_smaller_than_the_sphere_falls_back_to_the_whole_scene():
    """csrc/sphere_crop.hip sorts a host-known number of candidate slots (`cap`); a sphere holding more points than
    that is reported through *count and the query is repeated over the whole scene: same list either way."""
    g = np.load(GOLDEN)
    scene = SceneCropper(g["points"], in_radius=float(g["in_radius"]), num_points=700, device="cuda")
    pick, want = g["picks"][0], g["inds0"]
    assert len(want) > 64
    sorted_idx, count, _ = scene._sorted_scene(pick, cap=64)
    assert int(count) == len(want) and sorted_idx.numel() == 64          # reported, not truncated silently
    sorted_idx, count, _ = scene._sorted_scene(pick, cap=len(want))        # exactly enough
    assert np.array_equal(sorted_idx[:int(count)].cpu().numpy(), want)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4)
    s = scene._crop_native(pick, gen, cap=64)                              # too small -> repeated with cap = P
    inds = s["input_inds"].cpu().numpy()
    assert np.array_equal(np.sort(inds), np.sort(want[:700])) if len(want) >= 700 else np.isin(inds, want).all()
    #  the sample's random draws are made once: a retry with a larger sort capacity reuses them, so the
    # sample -- and the generator's state after it -- does not depend on the capacity the first attempt happened to have
    gen.manual_seed(4)
    direct = scene._crop_native(pick, gen, cap=scene.points64.shape[0])
    state_direct = gen.get_state().clone()
    gen.manual_seed(4)
    retried = scene._crop_native(pick, gen, cap=64)
    assert torch.equal(gen.get_state(), state_direct)
    assert torch.equal(retried["input_inds"], direct["input_inds"]) and torch.equal(retried["points"], direct["points"])

"""Vote bookkeeping and scene metrics: closerlook3d_amd/voting.py against the numpy
restatement of the reference's `validate` loop (oracle/voting.py, bit for bit) and against values produced by the
reference's own utils/util.py (tests/golden/voting_metrics.npz).  The same cases run on the CPU (torch) and,
marked gpu, on the device."""
import os

import numpy as np
import pytest
import torch

from closerlook3d_amd import voting
from oracle import voting as ov

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "voting_metrics.npz")
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def _batches(rng, sizes, C, B, N, steps):
    """Sphere-crop-like batches: distinct scene indices per element, wrap-around padding under mask 0."""
    for _ in range(steps):
        pred = rng.normal(size=(B, C, N)).astype(np.float32)
        mask = np.zeros((B, N), dtype=np.int32)
        inds = np.zeros((B, N), dtype=np.int64)
        label = rng.integers(0, len(sizes), size=B)
        if B > 1:
            label[1] = label[0]  # two elements of one batch in the same scene, overlapping
        for b in range(B):
            n = sizes[label[b]]
            valid = int(rng.integers(N // 2, N + 1)) if n >= N else n
            valid = min(valid, n)
            pick = rng.permutation(n)[:valid]
            inds[b, :valid] = pick
            inds[b, valid:] = pick[rng.integers(0, valid, size=N - valid)]
            mask[b, :valid] = 1
        yield pred, mask, inds, label


@pytest.mark.parametrize("device", DEVICES)
def test_vote_arrays_match_the_reference_loop_bit_for_bit(device):
    rng = np.random.default_rng(5)
    C, B, N = 13, 4, 600
    sizes = [5000, 700, 350]          # the last scene is smaller than a crop: every element pads
    arrays = ov.new_arrays(C, sizes)
    votes = voting.VoteAccumulator(C, sizes, test_smooth=0.95, device=device)
    for pred, mask, inds, label in _batches(rng, sizes, C, B, N, steps=12):
        ov.collect(arrays, pred, mask, inds, label, 0.95)
        votes.update(torch.from_numpy(pred).to(device), torch.from_numpy(mask).to(device),
                     torch.from_numpy(inds).to(device), label.tolist())
    sums, counts, vote_logits, running = arrays
    for c in range(len(sizes)):
        assert np.array_equal(votes.sums[c].cpu().numpy().view(np.uint32), sums[c].view(np.uint32))
        assert np.array_equal(votes.counts[c].cpu().numpy().view(np.uint32), counts[c].view(np.uint32))
        assert np.array_equal(votes.running[c].cpu().numpy().view(np.uint32), running[c].view(np.uint32))
        touched = counts[c][0] > 0.5   # the reference refreshes a scene's quotient only when it is voted on
        got = votes.vote_logits(c).cpu().numpy()
        assert np.array_equal(got[:, touched].view(np.uint32), vote_logits[c][:, touched].view(np.uint32))
        assert not got[:, ~touched].any()


@pytest.mark.parametrize("device", DEVICES)
def test_running_logits_carry_over_between_epochs(device):
    rng = np.random.default_rng(6)
    C, sizes = 5, [300, 200]
    arrays = ov.new_arrays(C, sizes)
    first = list(_batches(rng, sizes, C, 2, 100, steps=3))
    for pred, mask, inds, label in first:
        ov.collect(arrays, pred, mask, inds, label, 0.9)
    # second validation call: fresh sums / counts, the running logits of the first one
    carried = [r.copy() for r in arrays[3]]
    arrays2 = ov.new_arrays(C, sizes)
    arrays2 = (arrays2[0], arrays2[1], arrays2[2], [r.copy() for r in carried])
    votes = voting.VoteAccumulator(C, sizes, test_smooth=0.9, device=device, running=[torch.from_numpy(r) for r in carried])
    for pred, mask, inds, label in _batches(rng, sizes, C, 2, 100, steps=3):
        ov.collect(arrays2, pred, mask, inds, label, 0.9)
        votes.update(torch.from_numpy(pred), torch.from_numpy(mask), torch.from_numpy(inds), torch.from_numpy(label))
    for c in range(2):
        assert np.array_equal(votes.running[c].cpu().numpy().view(np.uint32), arrays2[3][c].view(np.uint32))


def test_oracle_metrics_match_the_reference_golden():
    g = np.load(GOLDEN)
    C = int(g["num_classes"])
    logits = [g[f"logits{i}"] for i in range(3)]
    iou, miou = ov.s3dis_metrics(C, logits, [g[f"proj{i}"] for i in range(3)], [g[f"labels{i}"] for i in range(3)])
    np.testing.assert_allclose(iou, g["iou"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(miou, g["miou"], rtol=0, atol=1e-12)
    siou, smiou = ov.sub_s3dis_metrics(C, logits, [g[f"sub_labels{i}"] for i in range(3)], g["prop"])
    np.testing.assert_allclose(siou, g["sub_iou"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(smiou, g["sub_miou"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(ov.iou_from_confusions(g["conf"]), g["conf_iou"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("device", DEVICES)
def test_metrics_match_the_reference_golden(device):
    """IoU_from_confusions, s3dis_metrics and sub_s3dis_metrics of the reference on seeded logits (a scene that never
    predicts four classes, two classes absent from the truth)."""
    g = np.load(GOLDEN)
    C = int(g["num_classes"])
    sizes = [g[f"logits{i}"].shape[1] for i in range(3)]
    votes = voting.VoteAccumulator(C, sizes, device=device)
    for i in range(3):  # one vote per point: the voted logits are the logits (x / (1 + 1e-6) keeps the arg-max)
        n = sizes[i]
        votes.update(torch.from_numpy(g[f"logits{i}"])[None], torch.ones(1, n, dtype=torch.int32),
                     torch.arange(n)[None], [i])
    iou, miou = voting.scene_metrics(votes, [g[f"proj{i}"] for i in range(3)], [g[f"labels{i}"] for i in range(3)])
    np.testing.assert_allclose(iou.cpu().numpy(), g["iou"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(float(miou), float(g["miou"]), rtol=0, atol=1e-9)
    siou, smiou = voting.sub_scene_metrics(votes, [g[f"sub_labels{i}"] for i in range(3)], g["prop"])
    np.testing.assert_allclose(siou.cpu().numpy(), g["sub_iou"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(float(smiou), float(g["sub_miou"]), rtol=0, atol=1e-6)
    got = voting.iou_from_confusions(torch.from_numpy(g["conf"]).to(device))
    np.testing.assert_allclose(got.cpu().numpy(), g["conf_iou"], rtol=0, atol=1e-9)


def test_confusion_matrix_drops_labels_outside_the_class_set():
    t = torch.tensor([0, 1, 2, 5, -1, 1])
    p = torch.tensor([0, 2, 2, 1, 0, 7])
    got = voting.confusion_matrix(t, p, 3).numpy()
    assert np.array_equal(got, ov.confusion(t.numpy(), p.numpy(), 3))
    assert got.sum() == 3

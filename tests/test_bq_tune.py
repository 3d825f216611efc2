"""Ball-query path by measurement (closerlook3d_amd/pt_utils.py: _bq_path): the keying of the
per-process table (CPU), and on the GPU box that the tuned call returns the library's bits and keeps the faster path."""
import numpy as np
import pytest
import torch

from closerlook3d_amd import pt_utils


def test_key_separates_sizes_devices_and_radius_buckets():
    k = pt_utils.bq_tune_key
    base = k(0, 16, 4096, 4096, 32, 0.141)
    assert base == k(0, 16, 4096, 4096, 32, 0.141)
    assert base == k(0, 16, 4096, 4096, 32, 0.145)          # same quarter-octave: the same stage with float noise in the radius
    assert base != k(0, 16, 4096, 4096, 32, 0.196)          # the dense variant (mean in-radius count 4 K instead of 1.5 K)
    assert base != k(0, 16, 4096, 4096, 32, 0.282)          # the next stage's radius (x 2)
    assert base != k(1, 16, 4096, 4096, 32, 0.141)          # another device
    assert base != k(0, 8, 4096, 4096, 32, 0.141) != k(0, 16, 1024, 4096, 32, 0.141) != k(0, 16, 4096, 4096, 16, 0.141)
    assert k(0, 1, 1, 1, 1, 0.0)[-1] is None and k(0, 1, 1, 1, 1, float("inf"))[-1] is None


def test_cpu_tensors_pinned_paths_and_small_problems_take_the_library_choice(monkeypatch):
    x = torch.zeros(16, 4096, 3)
    m = torch.ones(16, 4096, dtype=torch.int32)
    assert pt_utils._bq_path(x, x, m, m, 0.1, 32) == 0      # CPU tensors: the native op raises "CPU not supported" later
    monkeypatch.setattr(pt_utils, "BQ_TUNE", False)
    assert pt_utils._bq_path(x, x, m, m, 0.1, 32) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("mult", [1.5, 4.0])
def test_tuned_query_returns_the_same_bits_and_keeps_the_faster_path(mult, monkeypatch):
    from closerlook3d_amd import _ext
    from oracle import operators as oo
    monkeypatch.delenv("CL3D_BQ_PATH", raising=False)
    monkeypatch.setattr(pt_utils, "_BQ_PATH_TABLE", {})
    monkeypatch.setattr(pt_utils, "BQ_TUNE", True)
    B, N, K = 16, 4096, 32
    rng = np.random.default_rng(3)
    xyz_np, mask_np = oo.make_cloud(rng, B, N, pad_frac=0.1)
    radius = float((mult * K * 3 / (4 * np.pi * N)) ** (1 / 3))
    xyz, mask = torch.from_numpy(xyz_np).cuda(), torch.from_numpy(mask_np).cuda()
    want = _ext.masked_ordered_ball_query(xyz, xyz, mask, mask, radius, K)
    got = pt_utils._ball_query(xyz, xyz, mask, mask, radius, K)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    (key, (path, times)), = pt_utils._BQ_PATH_TABLE.items()
    assert key == pt_utils.bq_tune_key(0, B, N, N, K, radius)
    assert set(times) == {1, 2} and path == min(times, key=times.get), (path, times)
    # every named path gives the library's bits (the tuner may pick any of them)
    for p in (1, 2, 3):
        o = _ext.masked_ordered_ball_query(xyz, xyz, mask, mask, radius, K, path=p)
        assert torch.equal(o[0], want[0]) and torch.equal(o[1], want[1]), p
    # a second call is a table hit: no new entry, same answer
    again = pt_utils._ball_query(xyz, xyz, mask, mask, radius, K)
    assert len(pt_utils._BQ_PATH_TABLE) == 1 and torch.equal(again[0], want[0])
    # a path that does not take the sizes is refused, not replaced
    small = torch.zeros(1, 8, 3, device="cuda")
    sm = torch.ones(1, 8, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError):
        _ext.masked_ordered_ball_query(small, small, sm, sm, 0.1, 4, path=2)

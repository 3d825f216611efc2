"""GPU: pin the oracle AND the HIP engine to the reference's own kernels.

oracle/_ref/ref_ext.so is the reference's extension compiled for gfx950 by oracle/build_ref.py (its
own sources, PyTorch's stock ROCm extension toolchain).  It is test infrastructure; it is never
loaded by the product.  Skipped when the file is absent."""
import os

import numpy as np
import pytest
import torch

from oracle import build_ref
from oracle import native as on
from oracle import operators as oo
from tests.helpers import assert_close

pytestmark = pytest.mark.gpu

HAVE_REF = os.path.exists(os.path.join(build_ref.OUT_DIR, "ref_ext.so"))


@pytest.fixture(scope="module")
def ref():
    if not HAVE_REF:
        pytest.skip("oracle/_ref/ref_ext.so not built")
    return build_ref.load()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("B,N,K,mult,kind,pad", [(2, 512, 16, 1.5, "uniform", 0.2), (2, 1024, 16, 5.0, "planes", 0.1),
                                                (1, 4096, 32, 1.5, "uniform", 0.0), (1, 2048, 32, 4.0, "planes", 0.1)])
def test_ball_query_three_way(ref, B, N, K, mult, kind, pad):
    from closerlook3d_amd import _ext
    rng = np.random.default_rng(N + K)
    s, sm = oo.make_cloud(rng, B, N, kind=kind, pad_frac=pad)
    r = float((mult * K * 3 / (4 * np.pi * N)) ** (1 / 3))
    ri, rm = ref.masked_ordered_ball_query(_t(s), _t(s), _t(sm), _t(sm), r, K)
    oi, om = on.masked_ordered_ball_query(s, s, sm, sm, r, K)
    hi, hm = _ext.masked_ordered_ball_query(_t(s), _t(s), _t(sm), _t(sm), r, K)
    assert np.array_equal(ri.cpu().numpy(), oi), "oracle != reference kernel"
    assert np.array_equal(rm.cpu().numpy(), om)
    assert torch.equal(ri, hi), "HIP engine != reference kernel"
    assert torch.equal(rm, hm)


@pytest.mark.parametrize("B,N,m,dl,kind,pad", [(2, 256, 64, 0.12, "uniform", 0.25), (2, 1024, 1024, 0.2, "uniform", 0.1),
                                              (1, 4096, 1024, 0.04, "uniform", 0.0), (1, 1000, 300, 0.03, "planes", 0.0)])
def test_grid_subsampling_three_way(ref, B, N, m, dl, kind, pad):
    from closerlook3d_amd import _ext
    rng = np.random.default_rng(N + m)
    xyz, mask = oo.make_cloud(rng, B, N, kind=kind, pad_frac=pad)
    xyz = (xyz * 1.7 - 0.6).astype(np.float32)
    rs, rmask = ref.masked_grid_subsampling(_t(xyz), _t(mask), m, dl)
    os_, omask = on.masked_grid_subsampling(xyz, mask, m, dl)
    hs, hmask = _ext.masked_grid_subsampling(_t(xyz), _t(mask), m, dl)
    assert np.array_equal(rs.cpu().numpy().view(np.uint32), os_.view(np.uint32)), "oracle != reference kernel"
    assert np.array_equal(rmask.cpu().numpy(), omask)
    assert torch.equal(rs, hs) and torch.equal(rmask, hmask), "HIP engine != reference kernel"


def test_nearest_and_group_three_way(ref):
    from closerlook3d_amd import _ext
    rng = np.random.default_rng(77)
    B, N, M, K, C = 2, 1024, 300, 16, 8
    s, sm = oo.make_cloud(rng, B, N, pad_frac=0.2)
    q = rng.random((B, M, 3), dtype=np.float32)
    qm = (rng.random((B, M)) > 0.2).astype(np.int32)
    ri, rm = ref.masked_nearest_query(_t(q), _t(s), _t(qm), _t(sm))
    oi, om = on.masked_nearest_query(q, s, qm, sm)
    hi, hm = _ext.masked_nearest_query(_t(q), _t(s), _t(qm), _t(sm))
    assert np.array_equal(ri.cpu().numpy(), oi) and np.array_equal(rm.cpu().numpy(), om)
    assert torch.equal(ri, hi) and torch.equal(rm, hm)
    f = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (B, M, K)).astype(np.int32)
    rg = ref.group_points(_t(f), _t(idx))
    assert torch.equal(rg, _ext.group_points(_t(f), _t(idx)))
    assert np.array_equal(rg.cpu().numpy(), on.group_points(f, idx))
    g = rng.standard_normal((B, C, M, K)).astype(np.float32)
    rgg = ref.group_points_grad(_t(g), _t(idx), N).cpu().numpy()
    assert_close(_ext.group_points_grad(_t(g), _t(idx), N).cpu().numpy(), rgg, 1e-5, "hip vs ref scatter")
    assert_close(on.group_points_grad(g, idx, N), rgg, 1e-5, "oracle vs ref scatter")

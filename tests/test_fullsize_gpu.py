"""GPU, at BASELINE.json's full metric shape (B=16, N=M=4096, K=32, C=64/72): bit-exact checks against the CPU
oracle where it finishes in seconds, size-independent properties elsewhere, and fused-vs-grouped operator parity."""
import numpy as np
import pytest
import torch

from oracle import native as on
from oracle import operators as oo
from tests.helpers import assert_close, default_config

pytestmark = pytest.mark.gpu

B, N, K = 16, 4096, 32
RADIUS = float((1.5 * K * 3 / (4 * np.pi * N)) ** (1 / 3))


@pytest.fixture(scope="module")
def clouds():
    rng = np.random.default_rng(2024)
    xyz, mask = oo.make_cloud(rng, B, N, kind="uniform", pad_frac=0.0)
    x2, m2 = oo.make_cloud(rng, 4, N, kind="planes", pad_frac=0.1)  # last four clouds: surfaces, padded
    xyz[-4:], mask[-4:] = x2, m2
    return xyz, mask


def _dev(*a):
    return [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in a]


def test_ball_query_full_shape_bit_exact_and_properties(clouds):
    from closerlook3d_amd import _ext
    xyz, mask = clouds
    idx, msk = _ext.masked_ordered_ball_query(*_dev(xyz, xyz, mask, mask), RADIUS, K)
    want_idx, want_msk = on.masked_ordered_ball_query(xyz, xyz, mask, mask, RADIUS, K)
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    assert np.array_equal(msk.cpu().numpy(), want_msk)
    # properties that need no oracle
    x = torch.from_numpy(xyz).cuda()
    nb = torch.gather(x.unsqueeze(1).expand(B, N, N, 3), 2, idx.long().unsqueeze(-1).expand(B, N, K, 3))
    d2 = ((nb - x.unsqueeze(2)) ** 2).sum(-1)
    m = msk.bool()
    assert (d2[m] < RADIUS * RADIUS * (1 + 1e-5)).all(), "a reported neighbour lies outside the ball"
    valid_q = torch.from_numpy(mask).cuda().bool()
    cnt = msk.sum(-1)
    prefix = (torch.arange(K, device="cuda")[None, None, :] < cnt[..., None])
    assert torch.equal(m, prefix & valid_q[..., None] | (m & ~valid_q[..., None])), "mask must be a prefix of ones"
    inc = d2[..., 1:] >= d2[..., :-1] - 1e-7
    both = m[..., 1:] & m[..., :-1]
    assert inc[both].all(), "neighbours must come in non-decreasing distance"
    assert (idx[..., 0] == torch.arange(N, device="cuda")[None, :])[valid_q].float().mean() > 0.99, \
        "a valid query of the support set is (almost always) its own nearest neighbour"


def test_grid_subsampling_and_nearest_full_shape_bit_exact(clouds):
    from closerlook3d_amd import _ext
    xyz, mask = clouds
    sub, sm = _ext.masked_grid_subsampling(*_dev(xyz, mask), 1024, 0.04)
    want = on.masked_grid_subsampling(xyz, mask, 1024, 0.04)
    assert np.array_equal(sm.cpu().numpy(), want[1])
    assert np.array_equal(sub.cpu().numpy().view(np.uint32), want[0].view(np.uint32))
    ni, nm = _ext.masked_nearest_query(*_dev(xyz, want[0], mask, want[1]))
    wi, wm = on.masked_nearest_query(xyz, want[0], mask, want[1])
    assert np.array_equal(ni.cpu().numpy(), wi) and np.array_equal(nm.cpu().numpy(), wm)
    # idempotence-type property: subsampling the barycentres again with the same cell size keeps every one of them
    again, am = _ext.masked_grid_subsampling(sub, sm, 1024, 0.04 * 1e-3)
    assert int(am.sum()) == int(sm.sum())


def test_group_round_trip_full_shape(clouds):
    """gather then scatter: scatter(ones) counts references exactly; scatter is linear; gather is a pure copy."""
    from closerlook3d_amd import _ext
    xyz, mask = clouds
    C = 64
    idx, _ = _ext.masked_ordered_ball_query(*_dev(xyz, xyz, mask, mask), RADIUS, K)
    f = torch.randn(B, C, N, device="cuda")
    g = _ext.group_points(f, idx)
    b_i = torch.randint(0, B, (64,)); c_i = torch.randint(0, C, (64,)); j_i = torch.randint(0, N, (64,)); k_i = torch.randint(0, K, (64,))
    assert torch.equal(g[b_i, c_i, j_i, k_i], f[b_i, c_i, idx[b_i, j_i, k_i].long()])
    ones = torch.ones(B, 1, N, K, device="cuda")
    cnt = _ext.group_points_grad(ones, idx, N)[:, 0]
    want = torch.stack([torch.bincount(idx[b].flatten().long(), minlength=N) for b in range(B)]).float()
    assert torch.equal(cnt, want)
    a, c = torch.randn(B, C, N, K, device="cuda"), torch.randn(B, C, N, K, device="cuda")
    lhs = _ext.group_points_grad(a + c, idx, N)
    rhs = _ext.group_points_grad(a, idx, N) + _ext.group_points_grad(c, idx, N)
    assert_close(lhs.cpu().numpy(), rhs.cpu().numpy(), 1e-5, "scatter linearity")
    # <gather(f), a> == <f, scatter(a)>  (adjointness), in double
    lhs2 = (g.double() * a.double()).sum()
    rhs2 = (f.double() * _ext.group_points_grad(a, idx, N).double()).sum()
    # (relative to the size of the terms: the sum itself can land near zero for an unlucky draw)
    assert abs(lhs2 - rhs2) / (g.double() * a.double()).abs().sum() < 1e-7


@pytest.mark.parametrize("kind,over,C", [
    ("pospool", {"pospool__position_embedding": "xyz", "pospool__reduction": "avg"}, 72),
    ("adaptive_weight", {"adaptive_weight__num_mlps": 1, "adaptive_weight__reduction": "avg"}, 64),
    ("pointwisemlp", {"pointwisemlp__feature_type": "dp_fi_df", "pointwisemlp__num_mlps": 1, "pointwisemlp__reduction": "max"}, 64),
    ("pseudo_grid", {"pseudo_grid__KP_influence": "linear"}, 64),
])
def test_operators_full_shape_fused_vs_grouped(clouds, kind, over, C):
    """Both implementations on the GPU at the benchmark shape: same outputs (1e-5) and gradients."""
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    xyz, mask = clouds
    t_xyz, t_mask = _dev(xyz, mask)
    torch.manual_seed(11)
    feats = torch.randn(B, C, N, device="cuda")
    probe = torch.randn(B, C, N, device="cuda")
    res = {}
    for impl in ("fused", "grouped"):
        torch.manual_seed(3)
        mod = LocalAggregation(C, C, RADIUS, K, default_config(kind, over, cl3d_impl=impl)).cuda().train(True)
        f = feats.clone().requires_grad_(True)
        out = mod(t_xyz, t_xyz, t_mask, t_mask, f)
        (out * probe).sum().backward()
        res[impl] = (out.detach(), f.grad, {k: p.grad for k, p in mod.named_parameters() if p.grad is not None})
    assert_close(res["fused"][0].cpu().numpy(), res["grouped"][0].cpu().numpy(), 1e-5, f"{kind} out")
    # The module ends in BatchNorm + ReLU (and PointWiseMLP in a max): an output that differs by one rounding can
    # sit on the other side of the ReLU threshold, which reroutes that element's whole upstream gradient to (up
    # to K) support points.  Such flips are legitimate and rare; everything else must agree tightly.
    gf, gg = res["fused"][1], res["grouped"][1]
    diff = (gf - gg).abs()
    bad = diff > 1e-5 * (1.0 + gg.abs())
    assert int(bad.sum()) <= 16 * K, f"{kind}: {int(bad.sum())} feature-gradient elements disagree"
    rel = ((gf - gg)[~bad].double().norm() / gg.double().norm()).item()
    assert rel < 1e-5, f"{kind}: relative L2 error of the feature gradient {rel:.2e}"
    # parameter gradients: a rerouted element (see above) lands in ONE output channel's row; allow two such rows
    for k, v in res["grouped"][2].items():
        d = (res["fused"][2][k] - v).double().reshape(v.shape[0], -1)
        scale = v.double().norm() / np.sqrt(v.shape[0]) + 1e-30   # typical row norm
        row_err = d.norm(dim=1) / scale
        nbad = int((row_err > 1e-4).sum())
        assert nbad <= 2, f"{kind}: parameter gradient {k}: {nbad} rows off, worst {row_err.max().item():.2e}"

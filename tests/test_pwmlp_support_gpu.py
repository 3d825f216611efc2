"""The PointWiseMLP's support-major backward pass (csrc/fused_pwmlp.hip, pwmlp_support_kernel, through the C ABI's
cl3d_pwmlp_bwd_support) against a float64 reading of its definition.  Semantics: the reference's autograd through
local_aggregation_operators.py:288-301 -- slot 0 of a query's neighbour list is its centre (:290) -- with the BatchNorm
backward folded into dy = A dz [arg-max] + Bc + D y (csrc/fused_pwmlp.hip header):

    dG_i = D (W_r . sum_s rel_s + sum_s H[centre_s] + |S_i| G_i) + |S_i| Bc + A hit_i      over the slots s = (j, k) -> i
    dH_i = sum over the queries j centred on i (idx[j, 0] == i) of  D sy_j + K Bc + A dz_j

Cases: duplicated points (several queries centred on one point, centres that are not the query itself), lists longer than
two rounds of a lane group (the in-line fetch), M != N, nsample not a multiple of four, channel counts whose lane
groups are not the 16-lane DPP rows (Co = 36, 10, 72: the LDS group sum) and more than one channel chunk (Co = 256).
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _cloud(B, N, M, seed, dup=False, spread=1.0):
    g = torch.Generator().manual_seed(seed)
    s = (torch.rand(B, N, 3, generator=g) * spread).cuda()
    if dup:  # clusters of identical points: many queries share a centre, several are centred on the same point
        s[:, N // 2:] = s[:, : N - N // 2]
    q = s[:, :M].contiguous() if M <= N else torch.cat([s, s[:, : M - N] + 0.01], 1).contiguous()
    return q, s.contiguous()


CASES = [  # B, N, M, K, radius, dup
    (2, 512, 512, 32, 0.25, False),
    (3, 300, 300, 16, 0.3, True),
    (2, 1000, 250, 26, 0.2, False),
    (1, 256, 256, 48, 2.0, False),   # every point in every ball: lists of 48+ slots, one centre for all
    (2, 200, 333, 7, 0.25, True),
    (8, 1024, 1024, 32, 0.15, False),  # B a multiple of 8: the XCD-aware tile order
]


def _reference(ght, wr, cA, cB, cD, hit, dz_t, sy, idx, q, s, radius, K):
    """float64, straight from the definition (index_add over all slots)."""
    B, N, two = ght.shape
    Co = two // 2
    M = idx.shape[1]
    d = torch.float64
    G, H = ght[..., :Co].to(d), ght[..., Co:].to(d)
    out = torch.zeros(B, N, 2 * Co, dtype=d, device=ght.device)
    inv_r = torch.tensor(1.0 / radius, dtype=torch.float32, device=ght.device)
    for b in range(B):
        flat = idx[b].reshape(-1).long()                                   # slot -> support point
        jj = torch.arange(M, device=ght.device).repeat_interleave(K)       # slot -> query
        centre = idx[b, :, 0].long()[jj]                                   # slot -> its query's centre
        rel = ((s[b][flat] - q[b][jj]) * inv_r).to(d)                      # the forward pass's float32 expression
        relsum = torch.zeros(N, 3, dtype=d, device=ght.device).index_add_(0, flat, rel)
        hsum = torch.zeros(N, Co, dtype=d, device=ght.device).index_add_(0, flat, H[b][centre])
        cnt = torch.bincount(flat, minlength=N).to(d)[:, None]
        ysum = relsum @ wr.to(d).t() + hsum + cnt * G[b]
        dG = cD.to(d) * ysum + cnt * cB.to(d) + cA.to(d) * hit[b].to(d).t()
        cj = idx[b, :, 0].long()
        per_q = cD.to(d) * sy[b].to(d) + K * cB.to(d) + cA.to(d) * dz_t[b].to(d)
        dH = torch.zeros(N, Co, dtype=d, device=ght.device).index_add_(0, cj, per_q)
        out[b, :, :Co], out[b, :, Co:] = dG, dH
    return out


@pytest.mark.parametrize("Co", [64, 36, 10, 72, 144, 256])
@pytest.mark.parametrize("B,N,M,K,radius,dup", CASES)
def test_support_pass_matches_its_definition(B, N, M, K, radius, dup, Co):
    from closerlook3d_amd import _ext, _lib, fused
    lib = _lib.lib()
    q, s = _cloud(B, N, M, seed=7 * N + K, dup=dup)
    qm = torch.ones(B, M, dtype=torch.int32, device="cuda")
    sm = torch.ones(B, N, dtype=torch.int32, device="cuda")
    idx, _ = _ext.masked_ordered_ball_query(q, s, qm, sm, radius, K)
    off, slots = fused.inverse_index(idx, N)
    g = torch.Generator(device="cuda").manual_seed(Co + N)
    rnd = lambda *shape: torch.randn(*shape, device="cuda", generator=g)
    ght, wr = rnd(B, N, 2 * Co), rnd(Co, 3)
    cA, cB, cD = rnd(Co), rnd(Co) * 0.1, rnd(Co) * 0.1
    hit, dz_t, sy = rnd(B, Co, N), rnd(B, M, Co), rnd(B, M, Co)
    qtab = torch.cat([q, idx[:, :, :1].contiguous().view(torch.float32)], 2).contiguous()  # what cl3d_pwmlp_bwd_rows leaves
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    got = torch.empty(B, N, 2 * Co, device="cuda")
    runs = []
    for _ in range(2):
        got.fill_(float("nan"))
        with _lib.on_device(ght.device):
            _lib.check(lib.cl3d_pwmlp_bwd_support(_p(ght), _p(wr), _p(cA), _p(cB), _p(cD), _p(hit), _p(dz_t), _p(sy), _p(qtab),
                                                  _p(s), float(radius), _p(off), _p(slots), B, N, M, K, Co, _p(got), st))
        torch.cuda.synchronize()
        runs.append(got.clone())
    assert torch.equal(runs[0], runs[1]), "the pass must give the same bits on every run"
    want = _reference(ght, wr, cA, cB, cD, hit, dz_t, sy, idx, q, s, radius, K)
    scale = float(want.abs().max())
    err = float((got.double() - want).abs().max()) / scale
    assert err <= 2e-6, err  # float32 sums of a list's rows against float64: rounding only


@pytest.mark.parametrize("strided", [False, True])
def test_operator_gradients_through_the_support_pass_match_the_grouped_dataflow(strided):
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    from tests.helpers import default_config
    torch.manual_seed(5)
    B, N, C = 4, 1024, 36
    M = N // 4 if strided else N
    s = torch.rand(B, N, 3, device="cuda")
    q = s[:, :M].contiguous()
    qm = torch.ones(B, M, dtype=torch.int32, device="cuda")
    sm = torch.ones(B, N, dtype=torch.int32, device="cuda")
    feats = torch.randn(B, C, N, device="cuda")
    probe = torch.randn(B, C, M, device="cuda")
    grads = []
    for impl in ("fused", "grouped"):
        torch.manual_seed(7)
        cfg = default_config("pointwisemlp", {"pointwisemlp__feature_type": "dp_fi_df"}, cl3d_impl=impl)
        la = LocalAggregation(C, C, 0.15, 24, cfg).cuda().train()
        f = feats.clone().requires_grad_(True)
        out = la(q, s, qm, sm, f)
        (out * probe).sum().backward()
        grads.append([f.grad.clone()] + [p.grad.clone() for p in la.parameters()])
    for a, b in zip(*grads):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) / scale <= 2e-4

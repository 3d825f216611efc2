"""The PointWiseMLP's support-major backward pass on a summary of the CSR slot lists (csrc/fused_pwmlp.hip,
pwmlp_summary_kernel / pwmlp_support_sum_kernel; semantics: reference local_aggregation_operators.py:288-301 through
autograd -- slot 0 of a query's neighbour list is its centre, :290).

1. the summary itself against a plain numpy reading of idx: per support point the summed relative positions and, per CSR
   position, the entry the pass reads -- the centre of a k != 0 slot's query, the flagged query id of a k == 0 slot --
   whether the CSR build's fill pass left the entries (round 4) or the summary kernel wrote them;
2. d ght from the summary pass against the slot-by-slot pass (cl3d_pwmlp_bwd_support) on the same inputs, through the
   C ABI -- including duplicated points (several queries centred on one point), lists longer than one 64-slot round,
   M != N, nsample not a multiple of four and a channel count that is not a multiple of four;
3. the operator's gradients with the summary on and off (CL3D_PW_SUMMARY) agree.
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


def _geometry(q, s, radius, K, entries_from_build=True):
    from closerlook3d_amd import _lib, fused
    from closerlook3d_amd import _ext
    B, M, _ = q.shape
    N = s.shape[1]
    qm = torch.ones(B, M, dtype=torch.int32, device="cuda")
    sm = torch.ones(B, N, dtype=torch.int32, device="cuda")
    idx, _ = _ext.masked_ordered_ball_query(q, s, qm, sm, radius, K)
    off, slots = fused.inverse_index(idx, N, entries=entries_from_build)
    assert (fused.inverse_entries(idx) is not None) == entries_from_build
    rec, ent = fused.support_summary(idx, N, q, s, radius)
    torch.cuda.synchronize()
    return idx, off, slots, rec, ent


CASES = [  # B, N, M, K, radius, dup
    (2, 512, 512, 32, 0.25, False),
    (3, 300, 300, 16, 0.3, True),
    (2, 1000, 250, 26, 0.2, False),
    (1, 256, 256, 48, 2.0, False),   # every point in every ball: lists of 48+ slots, one centre for all
    (2, 200, 333, 7, 0.25, True),
]


@pytest.mark.parametrize("from_build", [True, False])
@pytest.mark.parametrize("B,N,M,K,radius,dup", CASES)
def test_summary_matches_a_plain_reading_of_idx(B, N, M, K, radius, dup, from_build):
    q, s = _cloud(B, N, M, seed=N + K, dup=dup)
    idx, off, slots, rec, ent = _geometry(q, s, radius, K, from_build)
    idx_h, off_h = idx.cpu().numpy(), off.cpu().numpy()
    rec_h, ent_h = rec.cpu().numpy(), ent.cpu().numpy().view(np.uint32)
    rec_i = rec_h.view(np.int32)
    q_h, s_h = q.cpu().numpy().astype(np.float64), s.cpu().numpy().astype(np.float64)
    for b in range(B):
        lists = [[] for _ in range(N)]
        for j in range(M):
            for k in range(K):
                lists[idx_h[b, j, k]].append((j, k))
        for i in range(N):
            n = off_h[b, i + 1] - off_h[b, i]
            assert n == len(lists[i])
            s0, length = (int(x) for x in rec_i[b, i, 3:5])
            assert s0 == off_h[b, i] and length == n
            rel = sum((s_h[b, i] - q_h[b, j]) / radius for j, _ in lists[i]) if n else np.zeros(3)
            np.testing.assert_allclose(rec_h[b, i, :3], rel, rtol=0, atol=2e-5 * max(1, n))
            # CSR order = slot order = (j, k) order: the centre of a k != 0 slot's query, the flagged id of a k == 0 slot
            want = [(0x80000000 | j) if k == 0 else int(idx_h[b, j, 0]) for j, k in lists[i]]
            assert [int(e) for e in ent_h[b, s0: s0 + n]] == want, (b, i)


@pytest.mark.parametrize("Co", [64, 36, 10, 72, 144, 256])
@pytest.mark.parametrize("B,N,M,K,radius,dup", CASES)
def test_support_pass_on_the_summary_matches_the_slot_walk(B, N, M, K, radius, dup, Co):
    from closerlook3d_amd import _lib
    lib = _lib.lib()
    q, s = _cloud(B, N, M, seed=7 * N + K, dup=dup)
    idx, off, slots, rec, ent = _geometry(q, s, radius, K)
    g = torch.Generator(device="cuda").manual_seed(Co + N)
    rnd = lambda *shape: torch.randn(*shape, device="cuda", generator=g)
    ght, wr = rnd(B, N, 2 * Co), rnd(Co, 3)
    cA, cB, cD = rnd(Co), rnd(Co) * 0.1, rnd(Co) * 0.1
    hit, dz_t, sy = rnd(B, Co, N), rnd(B, M, Co), rnd(B, M, Co)
    qtab = torch.cat([q, idx[:, :, :1].contiguous().view(torch.float32)], 2).contiguous()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    old, new = torch.empty(B, N, 2 * Co, device="cuda"), torch.empty(B, N, 2 * Co, device="cuda")
    with _lib.on_device(ght.device):
        _lib.check(lib.cl3d_pwmlp_bwd_support(_p(ght), _p(wr), _p(cA), _p(cB), _p(cD), _p(hit), _p(dz_t), _p(sy), _p(qtab),
                                              _p(s), float(radius), _p(off), _p(slots), B, N, M, K, Co, _p(old), st))
        _lib.check(lib.cl3d_pwmlp_bwd_support_sum(_p(ght), _p(wr), _p(cA), _p(cB), _p(cD), _p(hit), _p(dz_t), _p(sy),
                                                  _p(rec), _p(ent), B, N, M, K, Co, _p(new), st))
    torch.cuda.synchronize()
    scale = float(old.abs().max())
    err = float((new - old).abs().max()) / scale
    assert err <= 2e-6, err  # count * row against repeated adds of the same rows: rounding only


@pytest.mark.parametrize("strided", [False, True])
def test_operator_gradients_with_and_without_the_summary(strided, monkeypatch):
    from closerlook3d_amd import fused
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
    cfg = default_config("pointwisemlp", {"pointwisemlp__feature_type": "dp_fi_df"}, cl3d_impl="fused")
    la = LocalAggregation(C, C, 0.15, 24, cfg).cuda().train()
    grads = []
    for on in (True, False):
        monkeypatch.setattr(fused, "SUPPORT_SUMMARY", on)
        la.zero_grad()
        f = feats.clone().requires_grad_(True)
        out = la(q, s, qm, sm, f)
        (out * probe).sum().backward()
        grads.append([f.grad.clone()] + [p.grad.clone() for p in la.parameters()])
    for a, b in zip(*grads):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) / scale <= 5e-6

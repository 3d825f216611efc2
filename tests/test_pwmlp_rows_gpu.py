"""The PointWiseMLP's two element-wise passes (csrc/fused_pwmlp.hip: cl3d_pwmlp_apply, cl3d_pwmlp_bwd_rows) against a
direct restatement, on shapes that take the whole-tile kernels (pwmlp_rows64_kernel: widths that are multiples of 64 on
clouds of a multiple of 64 queries, every load of a tile in flight at once) and on shapes that take the general kernel
(pwmlp_rows_kernel).  Semantics: reference local_aggregation_operators.py:288-301 under BatchNorm2d + ReLU + max --
out = ReLU(scale y* + shift); dz = upstream gradient where the arg-max's activation is positive; t* = idx[j, k*];
per-channel sums of dz, dz * xhat and dz * rel(k*) (double partials, summed here over the blocks' records); the query
table {coordinates, idx[j, 0]} of the support-major pass.  Everything except the sums is an exact copy / select, so those
are compared bit for bit.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # B, N, M, K, Co
    (2, 256, 256, 32, 64),     # rows64: one chunk
    (3, 512, 128, 16, 128),    # rows64: two chunks, M != N, K = 16
    (1, 640, 640, 64, 64),     # rows64: K = 64, every lane a slot
    (16, 1024, 1024, 32, 64),  # rows64: fewer tiles (256) than workgroups (1024 partial records)
    (20, 2048, 4096, 16, 64),  # rows64: 1280 tiles on 1024 workgroups -- the tile loop runs twice for a quarter of them
    (2, 256, 256, 32, 48),     # general kernel: width not a multiple of 64
    (2, 200, 200, 20, 64),     # general kernel: ragged tiles
    (2, 256, 192, 7, 72),      # general kernel
]


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _inputs(B, N, M, K, Co, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    support = torch.rand(B, N, 3, device="cuda", generator=g)
    query = support[:, :M].contiguous() if M <= N else torch.rand(B, M, 3, device="cuda", generator=g)
    idx = torch.randint(0, N, (B, M, K), device="cuda", generator=g, dtype=torch.int32)
    ystar = rnd(B, M, Co)
    kstar = torch.randint(0, K, (B, M, Co), device="cuda", generator=g, dtype=torch.uint8)
    gout = rnd(B, Co, M)
    scale, shift = rnd(Co), rnd(Co) * 0.3
    mean, invstd = rnd(Co) * 0.2, torch.rand(Co, device="cuda", generator=g) + 0.5
    return support, query, idx, ystar, kstar, gout, scale, shift, mean, invstd


@pytest.mark.parametrize("B,N,M,K,Co", SHAPES)
def test_apply_pass(B, N, M, K, Co):
    from closerlook3d_amd import _lib
    lib = _lib.lib()
    _, _, _, ystar, _, _, scale, shift, _, _ = _inputs(B, N, M, K, Co, seed=Co + M)
    out = torch.full((B, Co, M), float("nan"), device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    with _lib.on_device(ystar.device):
        _lib.check(lib.cl3d_pwmlp_apply(_p(ystar), _p(scale), _p(shift), B, M, Co, _p(out), st))
    torch.cuda.synchronize()
    # one fused multiply-add, then the ReLU; the float64 restatement rounds twice, so allow the last bit
    want = torch.relu((ystar.double() * scale.double() + shift.double()).float()).transpose(1, 2).contiguous()
    assert torch.allclose(out, want, rtol=2.5e-7, atol=0.0)


@pytest.mark.parametrize("B,N,M,K,Co", SHAPES)
def test_backward_rows_pass(B, N, M, K, Co):
    from closerlook3d_amd import _lib
    lib = _lib.lib()
    radius = 0.37
    support, query, idx, ystar, kstar, gout, scale, shift, mean, invstd = _inputs(B, N, M, K, Co, seed=3 * Co + M + K)
    n_partials = lib.cl3d_pwmlp_partials(B, M, Co)
    dz_cm = torch.full((B, Co, M), float("nan"), device="cuda")
    ts_cm = torch.full((B, Co, M), -1, device="cuda", dtype=torch.int32)
    dz_t = torch.full((B, M, Co), float("nan"), device="cuda")
    qtab = torch.full((B, M, 4), float("nan"), device="cuda")
    partial = torch.zeros(n_partials, Co, 8, device="cuda", dtype=torch.float64)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    runs = []
    for _ in range(2):
        with _lib.on_device(ystar.device):
            _lib.check(lib.cl3d_pwmlp_bwd_rows(_p(gout), 1, _p(ystar), _p(kstar), _p(idx), _p(query), _p(support), float(radius),
                                               _p(scale), _p(shift), _p(mean), _p(invstd), B, N, M, K, Co, _p(dz_cm), _p(ts_cm),
                                               _p(dz_t), _p(qtab), _p(partial), n_partials, st))
        torch.cuda.synchronize()
        runs.append(partial.clone())
    assert torch.equal(runs[0], runs[1]), "the partial sums must have the same bits on every run"

    z = (ystar.double() * scale.double() + shift.double()).float()
    dz = torch.where(z > 0, gout.transpose(1, 2), torch.zeros((), device="cuda"))          # [B,M,Co]
    ts = torch.gather(idx.long(), 2, kstar.long())                                             # [B,M,Co]
    assert torch.equal(dz_t, dz)
    assert torch.equal(dz_cm, dz.transpose(1, 2))
    assert torch.equal(ts_cm.long(), ts.transpose(1, 2))
    assert torch.equal(qtab[..., :3], query)
    assert torch.equal(qtab[..., 3].contiguous().view(torch.int32), idx[:, :, 0].contiguous())

    inv_r = (torch.ones((), dtype=torch.float32) / torch.tensor(radius, dtype=torch.float32)).cuda()  # 1.0f / radius
    sup_at = torch.gather(support[:, None].expand(B, M, N, 3), 2, ts[..., None].expand(B, M, Co, 3))  # [B,M,Co,3]
    rel = (sup_at - query[:, :, None, :]) * inv_r                                                       # float32, the pass's expression
    xhat = (ystar - mean) * invstd
    want = torch.stack([dz.double().sum((0, 1)), (dz * xhat).double().sum((0, 1)),
                        (dz * rel[..., 0]).double().sum((0, 1)), (dz * rel[..., 1]).double().sum((0, 1)),
                        (dz * rel[..., 2]).double().sum((0, 1))], 1)                                    # [Co,5]
    got = partial.sum(0)[:, :5]
    tol = 1e-12 * float(B * M) + 1e-9
    assert float((got - want).abs().max()) <= tol * max(1.0, float(want.abs().max())), float((got - want).abs().max())


@pytest.mark.parametrize("B,N,M,K,Co", [(2, 256, 256, 32, 64), (3, 512, 128, 16, 128), (2, 300, 200, 20, 70), (16, 4096, 4096, 32, 64)])
def test_hits_and_coefficients_in_one_launch_equal_the_two_launches(B, N, M, K, Co):
    """cl3d_pwmlp_bwd_hits_coeffs == cl3d_pwmlp_bn_backward_coeffs + cl3d_pwmlp_bwd_hits, bit for bit (the same device code
    per block, only the launch differs), incl. a channel count that is not a multiple of the four channels of a block."""
    from closerlook3d_amd import _lib
    lib = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(B + N + Co)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    n_partials = lib.cl3d_pwmlp_partials(B, M, Co)
    partial = rnd(n_partials, Co, 8).double()
    gamma, mean, invstd = rnd(Co), rnd(Co) * 0.2, torch.rand(Co, device="cuda", generator=g) + 0.5
    sums = rnd(Co, 6).double()
    dz_cm = rnd(B, Co, M) * (torch.rand(B, Co, M, device="cuda", generator=g) > 0.3)
    ts_cm = torch.randint(0, N, (B, Co, M), device="cuda", generator=g, dtype=torch.int32)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    count = float(B * M * K)
    out = []
    for merged in (False, True):
        coef = torch.full((5, Co), float("nan"), device="cuda")
        dwr = torch.full((Co, 3), float("nan"), device="cuda")
        hit = torch.full((B, Co, N), float("nan"), device="cuda")
        with _lib.on_device(hit.device):
            if merged:
                _lib.check(lib.cl3d_pwmlp_bwd_hits_coeffs(_p(partial), n_partials, count, _p(gamma), _p(mean), _p(invstd), _p(sums),
                                                          _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]), _p(coef[4]), _p(dwr),
                                                          _p(dz_cm), _p(ts_cm), B, N, M, Co, _p(hit), st))
            else:
                _lib.check(lib.cl3d_pwmlp_bn_backward_coeffs(_p(partial), n_partials, Co, count, _p(gamma), _p(mean), _p(invstd),
                                                             _p(sums), _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]),
                                                             _p(coef[4]), _p(dwr), st))
                _lib.check(lib.cl3d_pwmlp_bwd_hits(_p(dz_cm), _p(ts_cm), B, N, M, Co, _p(hit), st))
        torch.cuda.synchronize()
        out.append((coef, dwr, hit))
    for a, b in zip(*out):
        assert not torch.isnan(b).any()
        assert torch.equal(a, b)
    # and the scatter itself against index_add in double
    want = torch.zeros(B, Co, N, dtype=torch.float64, device="cuda")
    want.scatter_add_(2, ts_cm.long(), dz_cm.double())
    assert torch.allclose(out[1][2].double(), want, rtol=1e-6, atol=1e-6)

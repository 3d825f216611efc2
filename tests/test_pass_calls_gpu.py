"""One C-ABI call per pass (csrc/pass.hip, pass_calls._PointwiseMLPPass -- the eager caller's path; reference: one `_ext` call per
autograd node, pt_utils.py:16-61) against the kernel-by-kernel path: the same kernels on the same inputs, so outputs, every
gradient and the BatchNorm buffers must be BIT-equal -- over four training steps (the library captures a pass it sees twice
in a row with an identical argument block into a launch graph and replays it from then on: direct, captured and replayed
calls are all in the comparison), for M == N, a strided layer (M != N), padded clouds and a channel count whose lane groups
are not 16 wide; and the eager step through it must be repeatable, with and without the launch graphs."""
import ctypes

import numpy as np
import pytest
import torch

from tests.helpers import default_config

pytestmark = pytest.mark.gpu

CASES = [  # B, N, M, K, C, radius, pad
    (4, 1024, 1024, 16, 36, 0.15, 0.0),
    (2, 2048, 512, 32, 64, 0.12, 0.1),
    (3, 600, 600, 24, 72, 0.2, 0.1),
]


def _cloud(B, N, M, pad, seed):
    from oracle import operators as oo
    xyz, mask = oo.make_cloud(np.random.default_rng(seed), B, N, pad_frac=pad)
    s, sm = torch.from_numpy(xyz).cuda(), torch.from_numpy(mask).cuda()
    return s[:, :M].contiguous(), s, sm[:, :M].contiguous(), sm


@pytest.mark.parametrize("B,N,M,K,C,radius,pad", CASES)
def test_pass_calls_equal_the_kernel_by_kernel_path(B, N, M, K, C, radius, pad, monkeypatch):
    from closerlook3d_amd import fused
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    q, s, qm, sm = _cloud(B, N, M, pad, seed=N + K)
    torch.manual_seed(1)
    feats = torch.randn(B, C, N, device="cuda")
    probe = torch.randn(B, C, M, device="cuda")
    res = {}
    for on in (True, False):
        monkeypatch.setattr(fused, "PASS_CALLS", on)
        torch.manual_seed(2)
        cfg = default_config("pointwisemlp", {"pointwisemlp__feature_type": "dp_fi_df"}, cl3d_impl="fused")
        la = LocalAggregation(C, C, radius, K, cfg).cuda().train()
        steps = []
        for step in range(4):
            la.zero_grad(set_to_none=True)
            f = feats.clone().requires_grad_(True)
            out = la(q, s, qm, sm, f)
            took_pass = type(out.grad_fn).__name__.startswith("_PointwiseMLPPass")
            assert took_pass == on
            (out * probe).sum().backward()
            steps.append([out.detach().clone(), f.grad.clone()] + [p.grad.clone() for p in la.parameters()]
                         + [b.clone() for b in la.buffers()])
        torch.cuda.synchronize()
        res[on] = steps
    for a_step, b_step in zip(res[True], res[False]):
        assert len(a_step) == len(b_step)
        for a, b in zip(a_step, b_step):
            assert torch.equal(a, b), "the pass calls run the kernel-by-kernel path's kernels: the bits must agree"


def _graph_stats():
    from closerlook3d_amd import _lib
    c, r = ctypes.c_longlong(0), ctypes.c_longlong(0)
    _lib.check(_lib.lib().cl3d_pwmlp_pass_graph_stats(ctypes.byref(c), ctypes.byref(r)))
    return c.value, r.value


@pytest.mark.parametrize("graphs", [1, 0])
def test_pass_calls_are_repeatable_and_leave_nothing_in_flight(graphs):
    """Twenty eager steps through the pass calls with the buffers of every step dropped at once (the allocator hands their
    memory to the next step): every step's gradients equal the first step's, bit for bit -- a fork still writing into a
    released buffer would show up here."""
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    B, N, K, C = 8, 2048, 32, 64
    q, s, qm, sm = _cloud(B, N, N, 0.05, seed=3)
    torch.manual_seed(4)
    feats = torch.randn(B, C, N, device="cuda")
    probe = torch.randn(B, C, N, device="cuda")
    cfg = default_config("pointwisemlp", {"pointwisemlp__feature_type": "dp_fi_df"}, cl3d_impl="fused")
    la = LocalAggregation(C, C, 0.1, K, cfg).cuda().train()
    from closerlook3d_amd import _lib
    was = _lib.lib().cl3d_pwmlp_pass_graphs(graphs)
    before = _graph_stats()
    first = None
    for step in range(20):
        la.zero_grad(set_to_none=True)
        f = feats.clone().requires_grad_(True)
        out = la(q, s, qm, sm, f)
        (out * probe).sum().backward()
        # bit checksums, nothing kept on the device between steps: the loop settles into the allocator's steady state
        got = [int(t.view(torch.int32).long().sum()) for t in [f.grad] + [p.grad for p in la.parameters()]]
        del out, f
        if first is None:
            first = got
        assert got == first, f"step {step} differs from step 0"
    torch.cuda.synchronize()
    after = _graph_stats()
    _lib.lib().cl3d_pwmlp_pass_graphs(was)
    if graphs:  # the steady state of an eager loop: the allocator repeats its addresses, the passes are replayed
        assert after[0] - before[0] >= 2 and after[1] - before[1] >= 10, (before, after)
    else:
        assert after == before


def test_more_distinct_passes_than_graph_slots_fall_back_to_direct_launches():
    """Ten operators of ten shapes called round-robin: more distinct argument blocks than the library keeps launch graphs
    for (eight per direction).  Every round gives the first round's bits, and the library does not re-capture a graph per
    call (a graph is only replaced when it has been idle for 64 calls): at most 8 captures per direction."""
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    cfg = default_config("pointwisemlp", {"pointwisemlp__feature_type": "dp_fi_df"}, cl3d_impl="fused")
    ops = []
    for k in range(10):
        B, N, K, C = 2, 256 + 64 * k, 16, 32
        q, s, qm, sm = _cloud(B, N, N, 0.0, seed=20 + k)
        torch.manual_seed(k)
        la = LocalAggregation(C, C, 0.2, K, cfg).cuda().train()
        ops.append((la, q, s, qm, sm, torch.randn(B, C, N, device="cuda"), torch.randn(B, C, N, device="cuda")))
    before = _graph_stats()
    first = {}
    for rnd in range(6):
        for k, (la, q, s, qm, sm, feats, probe) in enumerate(ops):
            la.zero_grad(set_to_none=True)
            f = feats.clone().requires_grad_(True)
            out = la(q, s, qm, sm, f)
            (out * probe).sum().backward()
            got = [int(t.view(torch.int32).long().sum()) for t in [out.detach(), f.grad] + [p.grad for p in la.parameters()]]
            del out, f
            assert first.setdefault(k, got) == got, f"operator {k}, round {rnd}"
    torch.cuda.synchronize()
    after = _graph_stats()
    assert after[0] - before[0] <= 16, (before, after)


def test_a_retained_graph_can_be_differentiated_twice():
    """backward(retain_graph=True) followed by a second backward: the node keeps the forward pass's buffers alive, the second
    pass gives the first one's gradients (accumulated: twice the value, exactly -- x + x is exact)."""
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    B, N, K, C = 2, 512, 16, 32
    q, s, qm, sm = _cloud(B, N, N, 0.1, seed=9)
    torch.manual_seed(5)
    cfg = default_config("pointwisemlp", {"pointwisemlp__feature_type": "dp_fi_df"}, cl3d_impl="fused")
    la = LocalAggregation(C, C, 0.2, K, cfg).cuda().train()
    f = torch.randn(B, C, N, device="cuda").requires_grad_(True)
    probe = torch.randn(B, C, N, device="cuda")
    out = la(q, s, qm, sm, f)
    assert type(out.grad_fn).__name__.startswith("_PointwiseMLPPass")
    loss = (out * probe).sum()
    loss.backward(retain_graph=True)
    once = [f.grad.clone()] + [p.grad.clone() for p in la.parameters()]
    junk = [torch.randn(B, C, N, device="cuda") for _ in range(8)]  # (whatever the allocator hands out next)
    loss.backward()
    del junk
    for a, b in zip([f.grad] + [p.grad for p in la.parameters()], once):
        assert torch.equal(a, b + b)


REDUCE_OPS = [  # name, kind, overrides, C
    ("pospool_xyz", "pospool", {"pospool__position_embedding": "xyz", "pospool__reduction": "avg"}, 72),
    ("pospool_sincos", "pospool", {"pospool__position_embedding": "sin_cos", "pospool__reduction": "avg"}, 36),
    ("adaptive_weight", "adaptive_weight", {}, 64),
    ("adaptive_weight_shared", "adaptive_weight", {"adaptive_weight__shared_channels": 2, "adaptive_weight__reduction": "sum"}, 32),
    ("pseudo_grid", "pseudo_grid", {}, 64),
    ("pseudo_grid_constant", "pseudo_grid", {"pseudo_grid__KP_influence": "constant"}, 24),
]


@pytest.mark.parametrize("strided", [False, True])
@pytest.mark.parametrize("name,kind,over,C", REDUCE_OPS)
def test_reduce_pass_calls_equal_the_kernel_by_kernel_path(name, kind, over, C, strided, monkeypatch):
    """Round 6: cl3d_reduce_train_forward / _backward -- the one-call-per-pass path of PosPool /
    AdaptiveWeight / PseudoGrid (pass_calls._ReducePass) -- against fused._FusedReduce kernel by kernel: bit-equal outputs,
    feature and parameter gradients and BatchNorm buffers over four training steps (direct, captured and replayed calls),
    M == N and a strided layer, padded clouds; and the launch-graph cache is the one the PointWiseMLP passes use."""
    from closerlook3d_amd import fused
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    B, N, K, radius = 3, 1024, 16, 0.15
    M = 256 if strided else N
    q, s, qm, sm = _cloud(B, N, M, 0.1, seed=len(name) + C)
    torch.manual_seed(1)
    feats = torch.randn(B, C, N, device="cuda")
    probe = torch.randn(B, C, M, device="cuda")
    res = {}
    before = _graph_stats()
    for on in (True, False):
        monkeypatch.setattr(fused, "PASS_CALLS", on)
        torch.manual_seed(2)
        la = LocalAggregation(C, C, radius, K, default_config(kind, over, cl3d_impl="fused")).cuda().train()
        steps = []
        for step in range(4):
            la.zero_grad(set_to_none=True)
            f = feats.clone().requires_grad_(True)
            out = la(q, s, qm, sm, f)
            (out * probe).sum().backward()
            steps.append([out.detach().clone(), f.grad.clone()] + [p.grad.clone() for p in la.parameters() if p.grad is not None]
                         + [b.clone() for b in la.buffers()])
        torch.cuda.synchronize()
        res[on] = steps
        if on:
            mid = _graph_stats()
        else:
            assert _graph_stats() == mid  # the kernel-by-kernel path never touches the pass graphs
    for a_step, b_step in zip(res[True], res[False]):
        assert len(a_step) == len(b_step) and len(a_step) >= 4
        for a, b in zip(a_step, b_step):
            assert torch.equal(a, b), "the pass calls run the kernel-by-kernel path's kernels: the bits must agree"


@pytest.mark.parametrize("kind,over,C", [("pospool", {"pospool__position_embedding": "xyz", "pospool__reduction": "avg"}, 72),
                                         ("pseudo_grid", {}, 64)])
def test_reduce_pass_calls_are_replayed_in_an_eager_loop(kind, over, C):
    """Twenty eager steps through cl3d_reduce_train_forward / _backward with every step's buffers dropped at once: the
    allocator settles, the argument blocks repeat, the passes are captured once and replayed from then on (the counters
    they share with the PointWiseMLP passes say so), and every step's gradients equal the first step's bit for bit."""
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    B, N, K = 8, 2048, 32
    q, s, qm, sm = _cloud(B, N, N, 0.05, seed=5)
    torch.manual_seed(4)
    feats = torch.randn(B, C, N, device="cuda")
    probe = torch.randn(B, C, N, device="cuda")
    la = LocalAggregation(C, C, 0.1, K, default_config(kind, over, cl3d_impl="fused")).cuda().train()
    before = _graph_stats()
    first = None
    for step in range(20):
        la.zero_grad(set_to_none=True)
        f = feats.clone().requires_grad_(True)
        out = la(q, s, qm, sm, f)
        (out * probe).sum().backward()
        got = [int(t.view(torch.int32).long().sum()) for t in [f.grad] + [p.grad for p in la.parameters() if p.grad is not None]]
        del out, f
        if first is None:
            first = got
        assert got == first, f"step {step} differs from step 0"
    torch.cuda.synchronize()
    after = _graph_stats()
    assert after[0] - before[0] >= 2 and after[1] - before[1] >= 10, (before, after)


def test_reduce_pass_is_what_an_eager_step_takes_and_a_capture_does_not(monkeypatch):
    from closerlook3d_amd import fused
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    q, s, qm, sm = _cloud(2, 512, 512, 0.0, seed=9)
    la = LocalAggregation(24, 24, 0.2, 16, default_config("pospool", {"pospool__position_embedding": "xyz", "pospool__reduction": "avg"},
                                                          cl3d_impl="fused")).cuda().train()
    taken = []
    real = fused._reduce_pass
    monkeypatch.setattr(fused, "_reduce_pass", lambda *a, **k: (taken.append(1), real(*a, **k))[1])
    f = torch.randn(2, 24, 512, device="cuda", requires_grad=True)
    la(q, s, qm, sm, f).sum().backward()
    assert taken == [1]
    with torch.no_grad():  # no backward will follow: the kernel-by-kernel forward (nothing kept)
        la(q, s, qm, sm, f)
    assert taken == [1]

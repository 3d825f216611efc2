"""cl3d_fused_param_reduce (csrc/fused_reduce.hip, param_reduce_kernel): the per-block parameter-gradient partials of
AdaptiveWeight / PseudoGrid summed in double, against the same reduction written with torch in float64 -- the layouts
the reference's autograd leaves in conv0.weight.grad / conv0.bias.grad (local_aggregation_operators.py:188-214) and in
kernel_weights.grad (:383-419)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

OP_ADAPTIVE, OP_PSEUDOGRID = 2, 3


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("G,C,S", [(1024, 72, 4), (8, 36, 1), (333, 144, 8), (1, 12, 12), (4096, 64, 2)])
def test_adaptive_weight_partials(G, C, S):
    from closerlook3d_amd import _lib
    lib = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(G + C)
    # partial sums two orders of magnitude above their total, as in a real step
    dparam = torch.randn(G, C, 4, device="cuda", generator=g) * 100.0
    g0 = torch.empty(C // S, 3, device="cuda")
    g1 = torch.empty(C // S, device="cuda")
    with _lib.on_device(dparam.device):
        _lib.check(lib.cl3d_fused_param_reduce(OP_ADAPTIVE, _p(dparam), G, C, S, _p(g0), _p(g1),
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    want = dparam.double().sum(0).view(C // S, S, 4).sum(1)
    scale = float(dparam.abs().max()) * (G * S) ** 0.5
    assert float((g0.double() - want[:, :3]).abs().max()) <= 1e-6 * scale
    assert float((g1.double() - want[:, 3]).abs().max()) <= 1e-6 * scale
    # and exactly the float32 rounding of the float64 sum wherever no double rounding is in play
    assert torch.equal(g1, want[:, 3].float()) or float((g1 - want[:, 3].float()).abs().max()) <= 1e-7 * scale


@pytest.mark.parametrize("G,C,P", [(1024, 72, 15), (16, 36, 15), (257, 144, 16), (3, 10, 1)])
def test_pseudo_grid_partials(G, C, P):
    from closerlook3d_amd import _lib
    lib = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(G * 3 + C)
    dparam = torch.randn(G, C, 16, device="cuda", generator=g) * 10.0
    g1 = torch.full((P, C), float("nan"), device="cuda")
    with _lib.on_device(dparam.device):
        _lib.check(lib.cl3d_fused_param_reduce(OP_PSEUDOGRID, _p(dparam), G, C, P, None, _p(g1),
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    want = dparam.double().sum(0)[:, :P].t()
    scale = float(dparam.abs().max()) * G ** 0.5
    assert float((g1.double() - want).abs().max()) <= 1e-6 * scale


def test_bad_arguments_are_refused():
    from closerlook3d_amd import _lib
    lib = _lib.lib()
    d = torch.zeros(4, 12, 4, device="cuda")
    out = torch.zeros(12, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.cl3d_fused_param_reduce(0, _p(d), 4, 12, 4, _p(out), _p(out), st) != 0      # PosPool has no parameters
    assert lib.cl3d_fused_param_reduce(OP_ADAPTIVE, _p(d), 4, 12, 5, _p(out), _p(out), st) != 0  # 12 % 5
    assert lib.cl3d_fused_param_reduce(OP_PSEUDOGRID, _p(d), 4, 12, 17, None, _p(out), st) != 0  # > 16 kernel points

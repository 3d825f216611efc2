"""The hand-written MFMA contraction (csrc/mfma_gemm.hip) against float64 references.

f32 (v_mfma_f32_32x32x2_f32) must hold north_star's 1e-5; bf16 (v_mfma_f32_32x32x16_bf16, f32 accumulation) is
checked two ways: against a float64 product of the SAME bf16-rounded inputs (tight: proves the kernel) and against
the unrounded product at the declared bf16 tolerance (BF16_TOL, relative to the largest output magnitude).
Random, asymmetric operands: a transposed or mis-tiled result cannot pass.
Reference semantics: the PointWiseMLP's Conv2d (local_aggregation_operators.py:253-257,288-295) in the engine's
factored form, and the 1x1 Conv1d layers of the bottleneck (backbones/resnet.py:32-39,58-66).
"""
import pytest
import torch

from closerlook3d_amd import _lib

pytestmark = pytest.mark.gpu

TOL = 1e-5       # f32 path, relative to the largest magnitude of the result
BF16_TOL = 1e-2  # bf16 inputs (8-bit mantissa), f32 accumulation, K <= 2304, relative to the largest magnitude

SHAPES = [  # B, C, N, Co
    (2, 64, 4096, 64),     # the metric shape's operator
    (3, 72, 1000, 36),     # N not a multiple of the point tile, C not a multiple of the K chunk
    (1, 36, 250, 20),      # N % 4 != 0: the scalar staging fallback
    (2, 288, 512, 288),    # several output tiles and K chunks
    (1, 8, 64, 4),         # smaller than one tile in every direction
    (2, 36, 2048, 36),     # forward product: the LDS-free kernel (C in {32, 36, 64, 72}, whole 32-point blocks), 72 outputs = 3 column tiles
    (1, 72, 1024, 72),     # the same with 144 outputs: five column tiles over two workgroup rows
    (3, 32, 96, 16),       # one column tile, three blocks per cloud
]


def _dev():
    return torch.device("cuda:0")


def _p(t):
    return None if t is None else t.data_ptr()


def _st():
    return _lib.stream_ptr(_dev())


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _rel(got, want):
    return float((got.double() - want).abs().max() / want.abs().max().clamp_min(1e-30))


def _wcat64(W, Co, C):
    W = W.double()
    wc, wd = W[:, 3:3 + C], W[:, 3 + C:]
    return torch.cat([wd, wc - wd], 0)  # [2Co, C]


def _inputs(B, C, N, Co, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    f = torch.randn(B, C, N, generator=g).to(_dev())
    W = (torch.randn(Co, 3 + 2 * C, generator=g) / (C ** 0.5)).to(_dev())
    dght = torch.randn(B, N, 2 * Co, generator=g).to(_dev())
    dwr = torch.randn(Co, 3, generator=g).to(_dev())
    return f, W, dght, dwr


@pytest.mark.parametrize("B,C,N,Co", SHAPES)
@pytest.mark.parametrize("prec", [0, 1])
def test_point_gemm_forward_and_gradients(B, C, N, Co, prec):
    lib = _lib.lib()
    f, W, dght, dwr = _inputs(B, C, N, Co, seed=B * 1000 + C)
    wr = torch.empty(Co, 3, device=_dev())
    wcat = torch.empty(2 * Co, C, device=_dev())
    ght = torch.full((B, N, 2 * Co), float("nan"), device=_dev())
    ws_bytes = lib.cl3d_workspace_bytes(14, B, N, Co, 0, C)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=_dev())
    _lib.check(lib.cl3d_pwmlp_point_gemm_fwd(_p(f), _p(W), B, C, N, Co, prec, _p(ght), _p(wr), _p(wcat), _p(ws), ws_bytes, _st()))
    wc64 = _wcat64(W, Co, C)
    assert torch.equal(wr, W[:, :3].contiguous())
    assert torch.equal(wcat.double(), torch.cat([W[:, 3 + C:], W[:, 3:3 + C] - W[:, 3 + C:]], 0).double())  # f32 subtraction
    want = torch.einsum("bcn,oc->bno", f.double(), wc64)
    dfeat = torch.full((B, C, N), float("nan"), device=_dev())
    _lib.check(lib.cl3d_pwmlp_point_gemm_bwd_data(_p(dght), _p(wcat), B, C, N, Co, prec, _p(dfeat), _p(ws), ws_bytes, _st()))
    want_df = torch.einsum("bno,oc->bcn", dght.double(), wc64)
    dW = torch.full((Co, 3 + 2 * C), float("nan"), device=_dev())
    _lib.check(lib.cl3d_pwmlp_point_gemm_bwd_weight(_p(f), _p(dght), _p(dwr), B, C, N, Co, prec, _p(dW), _p(ws), ws_bytes, _st()))
    dwcat = torch.einsum("bno,bcn->oc", dght.double(), f.double())
    want_dW = torch.cat([dwr.double(), dwcat[Co:], dwcat[:Co] - dwcat[Co:]], 1)
    if prec == 0:
        assert _rel(ght, want) <= TOL
        assert _rel(dfeat, want_df) <= TOL
        assert _rel(dW, want_dW) <= 2e-5  # B*N-term sums
    else:
        # the kernel proper: same rounded inputs, exact products, f32 accumulation
        fr, wr_, gr = _bf16_round(f), _bf16_round(wcat), _bf16_round(dght)
        assert _rel(ght, torch.einsum("bcn,oc->bno", fr, wr_)) <= TOL
        assert _rel(dfeat, torch.einsum("bno,oc->bcn", gr, wr_)) <= TOL
        dwc_r = torch.einsum("bno,bcn->oc", gr, fr)
        assert _rel(dW[:, 3:], torch.cat([dwc_r[Co:], dwc_r[:Co] - dwc_r[Co:]], 1)) <= 2e-5
        # and the declared tolerance against the unrounded contraction
        assert _rel(ght, want) <= BF16_TOL
        assert _rel(dfeat, want_df) <= BF16_TOL
        assert _rel(dW, want_dW) <= BF16_TOL
    assert torch.equal(dW[:, :3], dwr)
    # fixed slice order: bit-identical run to run
    dW2 = torch.empty_like(dW)
    _lib.check(lib.cl3d_pwmlp_point_gemm_bwd_weight(_p(f), _p(dght), _p(dwr), B, C, N, Co, prec, _p(dW2), _p(ws), ws_bytes, _st()))
    assert torch.equal(dW, dW2)


# round 5: both gradients from one pass over d ght (pwmlp_point_grads_kernel).  (B, C, N, Co, fused?)
GRAD_SHAPES = [
    (16, 64, 4096, 64, 1),   # the metric shape: the unmasked (FULL) variant, 256 workgroups x 4 tiles
    (2, 64, 128, 64, 1),     # fewer tiles than workgroups
    (3, 36, 192, 20, 1),     # padded channels and columns (C = 36 of 64, 2 Co = 40 of 128)
    (1, 32, 64, 64, 1),      # a single tile
    (5, 64, 320, 32, 1),     # tiles_per_cloud = 5: the (cloud, tile) decode
    (2, 72, 256, 36, 0),     # C > 64: the call runs the two products one after the other
    (2, 64, 100, 64, 0),     # N not a multiple of the 64-point tile: likewise
]


@pytest.mark.parametrize("B,C,N,Co,fused", GRAD_SHAPES)
@pytest.mark.parametrize("pro", [False, True])
def test_point_gemm_both_gradients_in_one_call(B, C, N, Co, fused, pro):
    lib = _lib.lib()
    assert lib.cl3d_pwmlp_point_gemm_bwd_fused(B, C, N, Co, 0) == fused
    assert lib.cl3d_pwmlp_point_gemm_bwd_fused(B, C, N, Co, 1) == 0  # bf16: the two-product path
    f, W, dght, dwr = _inputs(B, C, N, Co, seed=B * 77 + C + N)
    g = torch.Generator(device="cpu").manual_seed(5)
    scale = (torch.rand(C, generator=g) + 0.5).to(_dev()) if pro else None
    shift = (torch.randn(C, generator=g) * 0.3).to(_dev()) if pro else None
    wr = torch.empty(Co, 3, device=_dev())
    wcat = torch.empty(2 * Co, C, device=_dev())
    ws_bytes = lib.cl3d_workspace_bytes(14, B, N, Co, 0, C)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=_dev())
    _lib.check(lib.cl3d_pwmlp_split_weight(_p(W), Co, C, _p(wr), _p(wcat), _st()))
    dfeat = torch.full((B, C, N), float("nan"), device=_dev())
    dW = torch.full((Co, 3 + 2 * C), float("nan"), device=_dev())
    _lib.check(lib.cl3d_pwmlp_point_gemm_bwd(_p(f), _p(scale), _p(shift), _p(dght), _p(wcat), _p(dwr), B, C, N, Co, 0,
                                             _p(dfeat), _p(dW), _p(ws), ws_bytes, _st()))
    wc64 = wcat.double()
    act = f.double() if not pro else torch.relu(torch.addcmul(shift[None, :, None], f, scale[None, :, None])).double()
    want_df = torch.einsum("bno,oc->bcn", dght.double(), wc64)
    dwcat = torch.einsum("bno,bcn->oc", dght.double(), act)
    want_dW = torch.cat([dwr.double(), dwcat[Co:], dwcat[:Co] - dwcat[Co:]], 1)
    assert _rel(dfeat, want_df) <= TOL
    assert _rel(dW, want_dW) <= 2e-5  # B*N-term sums
    assert torch.equal(dW[:, :3], dwr)
    # workgroup-ordered partial sums: bit-identical run to run, and each gradient alone equals the pair's
    dfeat2, dW2 = torch.full_like(dfeat, float("nan")), torch.full_like(dW, float("nan"))
    _lib.check(lib.cl3d_pwmlp_point_gemm_bwd(_p(f), _p(scale), _p(shift), _p(dght), _p(wcat), _p(dwr), B, C, N, Co, 0,
                                             _p(dfeat2), None, _p(ws), ws_bytes, _st()))
    _lib.check(lib.cl3d_pwmlp_point_gemm_bwd(_p(f), _p(scale), _p(shift), _p(dght), _p(wcat), _p(dwr), B, C, N, Co, 0,
                                             None, _p(dW2), _p(ws), ws_bytes, _st()))
    assert torch.equal(dfeat, dfeat2) and torch.equal(dW, dW2)
    # the separate entry points agree to rounding (another summation order)
    if not pro:
        dW3 = torch.empty_like(dW)
        _lib.check(lib.cl3d_pwmlp_point_gemm_bwd_weight(_p(f), _p(dght), _p(dwr), B, C, N, Co, 0, _p(dW3), _p(ws), ws_bytes, _st()))
        assert _rel(dW3, dW.double()) <= 2e-5


def test_point_gemm_both_gradients_argument_checks():
    lib = _lib.lib()
    x = torch.zeros(1, 64, 64, device=_dev())
    g = torch.zeros(1, 64, 128, device=_dev())
    w = torch.zeros(128, 64, device=_dev())
    out = torch.zeros(64, 131, device=_dev())
    assert lib.cl3d_pwmlp_point_gemm_bwd(_p(x), None, None, _p(g), _p(w), None, 1, 64, 64, 64, 0, None, None, None, 0, None) == -1
    assert lib.cl3d_pwmlp_point_gemm_bwd(_p(x), _p(x), None, _p(g), _p(w), None, 1, 64, 64, 64, 0, _p(x), None, None, 0, None) == -1
    assert lib.cl3d_pwmlp_point_gemm_bwd(_p(x), None, None, _p(g), _p(w), None, 1, 64, 64, 64, 0, None, _p(out), None, 0, None) == -3  # no scratch


@pytest.mark.parametrize("B,C,N,Co", [(2, 72, 4096, 144), (4, 144, 1000, 36), (1, 3, 130, 72), (2, 1152, 64, 576),
                                      (1, 288, 10000, 288), (16, 2304, 16, 1152), (16, 1152, 16, 2304)])  # deep: K slices
@pytest.mark.parametrize("prec", [0, 1])
def test_conv1x1_matches_conv1d(B, C, N, Co, prec):
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(C + Co)
    x = torch.randn(B, C, N, generator=g).to(_dev())
    W = (torch.randn(Co, C, generator=g) / (C ** 0.5)).to(_dev())
    dy = torch.randn(B, Co, N, generator=g).to(_dev())
    y = torch.full((B, Co, N), float("nan"), device=_dev())
    dx = torch.full((B, C, N), float("nan"), device=_dev())
    dW = torch.full((Co, C), float("nan"), device=_dev())
    ws_bytes = lib.cl3d_workspace_bytes(15, B, N, Co, 0, C)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=_dev())
    _lib.check(lib.cl3d_conv1x1_fwd(_p(x), _p(W), B, C, N, Co, prec, _p(y), _p(ws), ws_bytes, _st()))
    _lib.check(lib.cl3d_conv1x1_bwd_data(_p(dy), _p(W), B, C, N, Co, prec, _p(dx), _p(ws), ws_bytes, _st()))
    if prec == 0:  # without scratch the products run unsplit: same values up to the summation order
        y0 = torch.empty_like(y)
        _lib.check(lib.cl3d_conv1x1_fwd(_p(x), _p(W), B, C, N, Co, prec, _p(y0), None, 0, _st()))
        assert _rel(y0, y.double()) <= TOL
    _lib.check(lib.cl3d_conv1x1_bwd_weight(_p(x), _p(dy), B, C, N, Co, prec, _p(dW), _p(ws), ws_bytes, _st()))
    rnd = (lambda t: t.double()) if prec == 0 else _bf16_round
    xr, wr, gr = rnd(x), rnd(W), rnd(dy)
    assert _rel(y, torch.einsum("oc,bcn->bon", wr, xr)) <= TOL
    assert _rel(dx, torch.einsum("oc,bon->bcn", wr, gr)) <= TOL
    assert _rel(dW, torch.einsum("bon,bcn->oc", gr, xr)) <= 2e-5
    if prec == 1:
        assert _rel(y, torch.einsum("oc,bcn->bon", W.double(), x.double())) <= BF16_TOL
        assert _rel(dW, torch.einsum("bon,bcn->oc", dy.double(), x.double())) <= BF16_TOL
    else:  # the library's conv agrees too (what the bottleneck ran before)
        ref = torch.nn.functional.conv1d(x, W[:, :, None])
        assert _rel(y, ref.double()) <= 2e-5


@pytest.mark.parametrize("B,C,N,Co", [(16, 1152, 16, 2304), (16, 2304, 16, 1152), (2, 1152, 64, 576), (16, 576, 62, 288),
                                      (3, 1000, 37, 130)])
@pytest.mark.parametrize("prec", [0, 1])
def test_k_slices_summed_inside_the_launch(B, C, N, Co, prec):
    """Round 6: the K slices of a deep-stage product are summed by the workgroup that arrives last
    at each output tile, in slice order -- no second launch.  Held here: (a) the inference epilogue (per-row affine map,
    residual, ReLU: cl3d_conv1x1_bn_act_fwd) goes through that sum unchanged, against float64; (b) one bit pattern over
    40 launches, half of them racing a second product on another stream (arrival order differs, the sum's order does
    not), which also shows that every ticket counter is back at zero when a launch ends; (c) N = 62 / 37 (N % 4 != 0)
    takes the element-wise output path of the same sum."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(7 * C + Co)
    x = torch.randn(B, C, N, generator=g).to(_dev())
    W = (torch.randn(Co, C, generator=g) / (C ** 0.5)).to(_dev())
    scale = (0.5 + torch.rand(Co, generator=g)).to(_dev())
    shift = torch.randn(Co, generator=g).to(_dev())
    res = torch.randn(B, Co, N, generator=g).to(_dev())
    ws_bytes = lib.cl3d_workspace_bytes(15, B, N, Co, 0, C)
    assert ws_bytes > 0, "this shape was meant to be cut along K"
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=_dev())
    ws2 = torch.empty(ws_bytes, dtype=torch.uint8, device=_dev())
    rnd = (lambda t: t.double()) if prec == 0 else _bf16_round
    want = torch.relu(torch.einsum("oc,bcn->bon", rnd(W), rnd(x)) * scale.double()[None, :, None] + shift.double()[None, :, None]
                      + res.double())
    side = torch.cuda.Stream()
    y2 = torch.empty(B, Co, N, device=_dev())
    first = None
    for it in range(40):
        y = torch.full((B, Co, N), float("nan"), device=_dev())
        if it % 2:  # a second sliced product beside it on another stream: the two share the chip, tiles finish in another order
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _lib.check(lib.cl3d_conv1x1_fwd(_p(x), _p(W), B, C, N, Co, prec, _p(y2), _p(ws2), ws_bytes, side.cuda_stream))
        _lib.check(lib.cl3d_conv1x1_bn_act_fwd(_p(x), _p(W), _p(scale), _p(shift), _p(res), 1, B, C, N, Co, prec, _p(y),
                                               _p(ws), ws_bytes, _st()))
        torch.cuda.current_stream().wait_stream(side)
        if first is None:
            first = y.clone()
            assert _rel(y, want) <= 2 * TOL
        else:
            assert torch.equal(y, first), f"launch {it}: the sliced product is not bit-reproducible"
    plain = torch.einsum("oc,bcn->bon", rnd(W), rnd(x))
    assert _rel(y2, plain) <= TOL


def test_in_launch_sum_forced_on_every_tile_and_split_matches_the_two_launch_form():
    """The planner uses the in-launch slice sum only where it measured faster (64 x 64 tiles, <= 3 slices); the variant build
    of scripts/micro/gemm_plan_sweep.py (CL3D_GEMM_FORCE = tile and split, CL3D_GEMM_FUSED_SUM = which form) forces it --
    and the two-launch form -- onto the same product in a child process: same bits (both add the slices in order 0, 1, 2, ...),
    for 64 x 64 tiles with 2 and 3 slices, f32 and bf16, with the epilogue."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    var = os.path.join(root, "scripts", "micro", "var", "libcl3d_gemm_plan_env.so")
    if not os.path.exists(var):
        pytest.skip("variant library not built (python scripts/micro/gemm_plan_sweep.py --build; __graft_entry__.build() does)")
    import ctypes
    from closerlook3d_amd import _lib as _shipped
    handle = ctypes.CDLL(var)
    stale = [name for name in _shipped.SIGNATURES if not hasattr(handle, name)]
    if stale:  # (built before the ABI grew: the loader of the child process would refuse it)
        pytest.skip(f"variant library is older than the shipped one (lacks {stale[:3]}): rebuild it with __graft_entry__.build()")
    code = r"""
import os, sys, torch
sys.path.insert(0, %r)
from closerlook3d_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(5)
B, C, N, Co = 16, 1152, 16, 576
x = torch.randn(B, C, N, generator=g).to(dev); W = (torch.randn(Co, C, generator=g) / C ** 0.5).to(dev)
scale = (0.5 + torch.rand(Co, generator=g)).to(dev); shift = torch.randn(Co, generator=g).to(dev)
res = torch.randn(B, Co, N, generator=g).to(dev)
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
p = lambda t: t.data_ptr()
for prec in (0, 1):
    for split in (2, 3):
        outs = []
        for form in ("1", "0"):
            os.environ["CL3D_GEMM_FUSED_SUM"] = form
            os.environ["CL3D_GEMM_FORCE"] = "1,1,%%d" %% split
            y = torch.full((B, Co, N), float("nan"), device=dev)
            for _ in range(3):
                _lib.check(lib.cl3d_conv1x1_bn_act_fwd(p(x), p(W), p(scale), p(shift), p(res), 1, B, C, N, Co, prec, p(y), p(ws), ws.numel(), _lib.stream_ptr(dev)))
            torch.cuda.synchronize()
            outs.append(y)
        assert torch.equal(outs[0], outs[1]), (prec, split, float((outs[0] - outs[1]).abs().max()))
        want = torch.relu(torch.einsum("oc,bcn->bon", W.double(), x.double()) * scale.double()[None, :, None] + shift.double()[None, :, None] + res.double())
        err = float((outs[0].double() - want).abs().max() / want.abs().max())
        assert err <= (1e-5 if prec == 0 else 1e-2), (prec, split, err)
print("ok")
""" % root
    env = dict(os.environ, CL3D_LIB=var)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_bad_arguments_are_refused():
    lib = _lib.lib()
    assert lib.cl3d_pwmlp_point_gemm_fwd(None, None, 1, 8, 16, 4, 0, None, None, None, None, 0, None) == -1
    assert lib.cl3d_conv1x1_fwd(None, None, 1, 8, 16, 4, 7, None, None, 0, None) == -1  # precision 7
    x = torch.zeros(1, 8, 16, device=_dev())
    w = torch.zeros(4, 3 + 16, device=_dev())
    g = torch.zeros(1, 16, 8, device=_dev())
    # the PointWiseMLP weight gradient always goes through the reduce kernel: scratch is mandatory
    assert lib.cl3d_pwmlp_point_gemm_bwd_weight(_p(x), _p(g), None, 1, 8, 16, 4, 0, _p(w), None, 0, None) == -3


def test_measured_plans_give_the_model_plans_results_and_stay_put():
    """cl3d_gemm_autotune (round 6): with the switch on, the first eager call of a product times the plausible (tile, K
    split) plans and keeps one for the process.  Whatever it keeps, the result is the product (1e-5 relative in f32 against
    float64; the plans differ only in the order the K slices are added), repeated calls give the same bits (a key keeps
    its plan), a product first seen inside a stream capture takes the model's plan without synchronising, and the counters
    move."""
    import closerlook3d_amd
    from closerlook3d_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    p = lambda t: t.data_ptr()  # noqa: E731
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    was = closerlook3d_amd.gemm_autotune(True)
    try:
        before = closerlook3d_amd.gemm_autotune_stats()[0]
        for (B, C, N, Co) in ((16, 288, 256, 576), (16, 1152, 16, 1152), (4, 72, 4096, 144)):
            x = torch.randn(B, C, N, generator=g).to(dev)
            dy = torch.randn(B, Co, N, generator=g).to(dev)
            W = (torch.randn(Co, C, generator=g) / C ** 0.5).to(dev)
            y, dx, dW = torch.empty(B, Co, N, device=dev), torch.empty(B, C, N, device=dev), torch.empty(Co, C, device=dev)
            st = _lib.stream_ptr(dev)
            runs = []
            for _ in range(3):
                _lib.check(lib.cl3d_conv1x1_fwd(p(x), p(W), B, C, N, Co, 0, p(y), p(ws), ws.numel(), st))
                _lib.check(lib.cl3d_conv1x1_bwd_data(p(dy), p(W), B, C, N, Co, 0, p(dx), p(ws), ws.numel(), st))
                _lib.check(lib.cl3d_conv1x1_bwd_weight(p(x), p(dy), B, C, N, Co, 0, p(dW), p(ws), ws.numel(), st))
                torch.cuda.synchronize()
                runs.append((y.clone(), dx.clone(), dW.clone()))
            for a, b in zip(runs[0], runs[2]):
                assert torch.equal(a, b)
            want = (torch.einsum("oc,bcn->bon", W.double(), x.double()), torch.einsum("oc,bon->bcn", W.double(), dy.double()),
                    torch.einsum("bon,bcn->oc", dy.double(), x.double()))
            for got, ref in zip(runs[0], want):
                assert float((got.double() - ref).abs().max() / ref.abs().max()) <= 1e-5
        assert closerlook3d_amd.gemm_autotune_stats()[0] >= before + 9
        # an unseen product inside a capture: no measurement (it would have to synchronise), the model's plan, same product
        B, C, N, Co = 8, 144, 512, 200
        x = torch.randn(B, C, N, generator=g).to(dev)
        W = (torch.randn(Co, C, generator=g) / C ** 0.5).to(dev)
        y = torch.empty(B, Co, N, device=dev)
        seen = closerlook3d_amd.gemm_autotune_stats()[0]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            _lib.check(lib.cl3d_conv1x1_fwd(p(x), p(W), B, C, N, Co, 0, p(y), p(ws), ws.numel(), _lib.stream_ptr(dev)))
        assert closerlook3d_amd.gemm_autotune_stats()[0] == seen
        graph.replay()
        torch.cuda.synchronize()
        ref = torch.einsum("oc,bcn->bon", W.double(), x.double())
        assert float((y.double() - ref).abs().max() / ref.abs().max()) <= 1e-5
    finally:
        closerlook3d_amd.gemm_autotune(was)

"""A/B arms that run parts of the network through the VENDOR LIBRARY (torch.bmm -> rocBLAS / hipBLASLt, nn.Conv1d ->
MIOpen) or through the dataflow of earlier rounds, for timing against the engine.  They live OUTSIDE the product package
on purpose (`closerlook3d_amd/` has no vendor-library back end and no switch that selects one): a
script installs an arm by replacing functions of the imported package --

    from ab import library_arms
    library_arms.install(block="modules")      # scripts/bench_backbone.py --block modules
    library_arms.install(decode="cat")         # ... --decode cat
    library_arms.install(layerwise=True)       # ... --layerwise

-- and `PointRowsLibrary` is the library form of the PointWiseMLP's per-point contraction that
tests/test_operators_gpu.py::test_point_rows_weight_plumbing_matches_autograd[library] and scripts/bench_point_gemm.py
compare the engine's MFMA kernel with.
"""
import os
import sys

import torch
from torch.autograd import Function

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from closerlook3d_amd import _lib  # noqa: E402


def _p(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    return _lib.stream_ptr(t.device)


class PointRowsLibrary(Function):
    """The same contraction through the vendor library (torch.bmm -> rocBLAS / hipBLASLt) with the engine's two
    weight-plumbing kernels around it: the baseline of scripts/bench_point_gemm.py, not used by the operators."""

    @staticmethod
    def forward(ctx, features, W):
        B, C, N = features.shape
        Co = W.shape[0]
        W = W.contiguous()
        wr = torch.empty((Co, 3), dtype=torch.float32, device=W.device)
        wcat = torch.empty((2 * Co, C), dtype=torch.float32, device=W.device)
        with _lib.on_device(W.device):
            _lib.check(_lib.lib().cl3d_pwmlp_split_weight(_p(W), Co, C, _p(wr), _p(wcat), _stream(W)))
        ctx.save_for_backward(features, wcat)
        ght = torch.bmm(features.transpose(1, 2), wcat.t().unsqueeze(0).expand(B, -1, -1))
        return ght, wr

    @staticmethod
    def backward(ctx, dght, dwr):
        features, wcat = ctx.saved_tensors
        B, C, N = features.shape
        Co = wcat.shape[0] // 2
        dfeat = dW = None
        if dght is not None and ctx.needs_input_grad[0]:
            dfeat = torch.bmm(wcat.t().unsqueeze(0).expand(B, -1, -1), dght.transpose(1, 2))  # [B,C,N]
        if ctx.needs_input_grad[1]:
            dwb = (torch.bmm(features, dght) if dght is not None
                   else torch.zeros((B, C, 2 * Co), dtype=torch.float32, device=features.device))
            dW = torch.empty((Co, 3 + 2 * C), dtype=torch.float32, device=features.device)
            dwr = dwr.contiguous() if dwr is not None else None
            with _lib.on_device(features.device):
                _lib.check(_lib.lib().cl3d_pwmlp_merge_weight_grad(_p(dwr), _p(dwb), B, Co, C, _p(dW), _stream(dwb)))
        return dfeat, dW




def _run_conv_bn_modules(seq, x, impl='auto', precision='f32', residual=None, shortcut=None):
    """backbones.run_conv_bn module by module (nn.Conv1d, nn.BatchNorm1d, ReLU), as the reference and round 1 ran it."""
    y = seq(x)
    if residual is not None:
        y = torch.relu(y + (shortcut(residual) if shortcut is not None else residual))
    return y


def install(block=None, decode=None, layerwise=False):
    """Replace pieces of the imported engine (process-wide, for the rest of the run):
    block='modules'   every Conv1d + BatchNorm1d (+ ReLU) unit of the bottlenecks and decoders through the nn modules;
    decode='cat'      the segmentation decoders concatenate [up(f) ; skip] as the reference does;
    layerwise=True    PointWiseMLP bottlenecks layer by layer (the activated tensors between their layers in HBM)."""
    from closerlook3d_amd import backbones, fused
    if block == 'modules':
        backbones.run_conv_bn = _run_conv_bn_modules
        layerwise, decode = True, 'cat'
    if decode == 'cat':
        fused.decode_level = lambda *a, **k: None  # (the decoder then takes its concatenating path)
    if layerwise:
        backbones._FUSE_MIN_VALUES = 1 << 62

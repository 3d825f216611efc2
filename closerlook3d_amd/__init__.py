"""closerlook3d_amd -- MI355X-native local-aggregation engine behind CloserLook3D's operator API.

Layout (only what the hot path needs):
  csrc/                         hand-written HIP kernels for gfx950 + the C ABI (include/cl3d.h)
  build.py                      hipcc driver -> libcl3d.so (in-tree)
  _lib.py                       ctypes loader (fails loudly; no CPU fallback)
  _ext.py                       the reference's `pt_custom_ops._ext` function surface
  pt_utils.py                   the reference's grouping API (MaskedQueryAndGroup, ...)
  local_aggregation_operators   LocalAggregation / PosPool / AdaptiveWeight / PointWiseMLP / PseudoGrid
  fused.py                      fused-operator entry points
  backbones.py                  ResNet / Bottleneck / SceneSegHeadResNet callers (integration, bench)
  dp.py                         one-process-per-GPU sharding + RCCL gradient all-reduce
"""
__version__ = "0.1.0"


def whole_step_capture(on=True):
    """Context manager: the HIP graph being captured holds forward AND backward of every operator (see
    closerlook3d_amd.fused.whole_step_capture)."""
    from .fused import whole_step_capture as _w
    return _w(on)


def deferred_weight_gradients(on=True):
    """Context manager, inside a declared whole-step capture: the weight gradients of the engine's contractions stay on
    their side stream beside the rest of the backward pass instead of being joined layer by layer; the step calls
    join_weight_gradients() between backward() and whatever reads the parameters' .grad
    (closerlook3d_amd.fused.deferred_weight_gradients)."""
    from .fused import deferred_weight_gradients as _d
    return _d(on)


def join_weight_gradients():
    """Join the deferred weight gradients and hand each to its parameter's .grad (closerlook3d_amd.fused.join_weight_gradients)."""
    from .fused import join_weight_gradients as _j
    return _j()


def step_stream(device=None):
    """The HIP stream on which a training step should be BOTH warmed up and captured (one per device).

    torch.cuda.graph() asks for a warm-up on a side stream and then captures on another stream of its own.  Autograd runs
    every parameter's AccumulateGrad node on the stream that was current when the parameter was FIRST used -- the warm-up's
    -- so a step captured the usual way carries that stream as one more concurrent branch of its backward pass (event
    hand-overs around every weight gradient; with gradients accumulated in place, `p.grad += dW` kernels on it).  Round 6
    traced the replay-varying gradients of a captured backward pass with forked gradient products to exactly that branch
    (profiles/r06/two_graph_repeat_check.txt): with warm-up and capture on ONE stream the same layout is exact.

        s = closerlook3d_amd.step_stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step(); step()                                   # warm-up
        torch.cuda.current_stream().wait_stream(s)
        with closerlook3d_amd.whole_step_capture(), torch.cuda.graph(g, stream=s):
            step()
    """
    from .fused import step_stream as _s
    return _s(device)


def gemm_autotune(enable=True):
    """Tile / K-slice plans of the engine's dense products (csrc/mfma_gemm.hip: the 1x1 convolutions and the PointWiseMLP's
    per-point products) by MEASUREMENT instead of the launch-time model: process-wide, off by default; returns the previous
    setting.  On: the first eager call of a product (a warm-up step before a capture is enough) times the plausible plans
    on the caller's stream and keeps the winner for the process (`include/cl3d.h: cl3d_gemm_autotune`).  A plan fixes the
    order in which K slices are summed, so two processes may then differ in the last bits of a result; one process never
    does.  Measured on the config-2 backbone in DESIGN 3.3."""
    from . import _lib
    return bool(_lib.lib().cl3d_gemm_autotune(1 if enable else 0))


def gemm_autotune_stats():
    """(products measured so far, how many of them kept a plan other than the model's)."""
    import ctypes
    from . import _lib
    a, b = ctypes.c_longlong(0), ctypes.c_longlong(0)
    _lib.check(_lib.lib().cl3d_gemm_autotune_stats(ctypes.byref(a), ctypes.byref(b)))
    return int(a.value), int(b.value)

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

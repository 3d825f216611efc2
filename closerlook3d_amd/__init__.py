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

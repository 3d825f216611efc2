"""Fused local-aggregation kernels (no [B,C,M,K] materialisation) -- Python side.

`use_fused(impl, kind, module)` decides whether an operator instance takes the fused HIP path.
Until a kind is listed in `_AVAILABLE` the operators run their 'grouped' dataflow (still on the HIP
engine's native ops); `impl='fused'` on an unavailable kind raises instead of silently degrading.
"""
_AVAILABLE = set()


def use_fused(impl, kind, module):
    if impl == 'grouped':
        return False
    ok = kind in _AVAILABLE
    if impl == 'fused' and not ok:
        raise NotImplementedError(f"fused path for '{kind}' is not built in this version")
    return ok

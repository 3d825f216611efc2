"""ctypes loader for libcl3d.so -- the only way the Python layer reaches the HIP kernels.

There is no CPU fallback: if the shared object is missing or does not load, importing any op
raises.  (`closerlook3d_amd.build.build()` compiles it; `__graft_entry__.build()` calls that.)
"""
import ctypes
import os

import torch  # noqa: F401  -- imported first so libcl3d binds to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
_D2_FORM = int(os.environ.get("CL3D_D2_FORM", "0") or 0)  # build-time distance canon, see closerlook3d_amd/build.py
# CL3D_LIB: another build of the same library (scripts/micro/bq_variants.py times kernel variants through the whole engine)
_PATH = os.environ.get("CL3D_LIB") or os.path.join(_HERE, "libcl3d.so" if _D2_FORM == 0 else f"libcl3d_d2form{_D2_FORM}.so")
_lib = None

_P = ctypes.c_void_p  # device pointers travel as void*
_I = ctypes.c_int
_F = ctypes.c_float
_Z = ctypes.c_size_t

# name -> argtypes; restype is int for every op.  Mirrors include/cl3d.h (tests/test_abi.py checks
# that every symbol declared in the header is exported by the library and listed here).
SIGNATURES = {
    "cl3d_masked_ordered_ball_query": [_P, _P, _P, _P, _I, _I, _I, _F, _I, _P, _P, _P, _Z, _P],
    "cl3d_masked_ordered_ball_query_path": [_I, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P, _P, _P, _Z, _P],
    "cl3d_ball_query_paths": [_I, _I, _I],
    "cl3d_fused_supported": [_I, _I, _I],
    "cl3d_group_points": [_P, _P, _I, _I, _I, _I, _I, _P, _P],
    "cl3d_group_points_grad": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P],
    "cl3d_masked_grid_subsampling": [_P, _P, _I, _I, _I, _F, _P, _P, _P, _Z, _P],
    "cl3d_masked_nearest_query": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "cl3d_group_xyz_features": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P],
    "cl3d_build_inverse_index": [_P, _I, _I, _I, _P, _P, _P, _Z, _P],
    "cl3d_fused_reduce_fwd": [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _P, _P, _I, _F, _I, _P, _I, _P, _P, _P],
    "cl3d_fused_param_partials": [_I, _I, _I, _I],
    "cl3d_fused_param_reduce": [_I, _P, _I, _I, _I, _P, _P, _P],
    "cl3d_fused_reduce_bwd": [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _F, _I, _P, _I, _P, _I, _P],
    "cl3d_maxpool_fwd": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "cl3d_maxpool_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P],
    "cl3d_maxpool_fwd_targets": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "cl3d_maxpool_bwd_targets": [_P, _P, _I, _I, _I, _I, _P, _P],
    "cl3d_dataset_grid_subsampling": [_P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P, ctypes.c_size_t, _P],
    "cl3d_sphere_crop_query": [_P, _I, _P, ctypes.c_double, _I, _P, _P, _P, _Z, _P],
    "cl3d_sphere_crop_assemble": [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P],
    "cl3d_transpose": [_P, _I, _I, _I, _P, _P],
    "cl3d_transpose_bn_relu": [_P, _P, _P, _I, _I, _I, _P, _P],
    "cl3d_bn_rows_partials": [ctypes.c_longlong, _I],
    "cl3d_bn_rows_stats": [_P, ctypes.c_longlong, _I, _P, _I, ctypes.c_double, _F, _F] + [_P] * 10,
    "cl3d_bn_rows_bwd": [_P] * 7 + [ctypes.c_longlong, _I, ctypes.c_double, _P, _I, _P, _P, _P],
    "cl3d_bn_partials": [_I, _I, _I],
    "cl3d_bn_relu_stats": [_P, _I, _I, _I, _P, _I, ctypes.c_double, _F, _F] + [_P] * 10,
    "cl3d_bn_relu_apply": [_P, _P, _P, _I, _I, _I, _P, _P],
    "cl3d_bn_relu_bwd": [_P] * 7 + [_I, _I, _I, ctypes.c_double, _P, _I, _P, _P, _P],
    "cl3d_bn_add_relu_apply": [_P] * 6 + [_I, _I, _I, _I, _P, _P],
    "cl3d_bn_add_relu_train_fwd": [_P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P],
    "cl3d_bn_add_relu_bwd": [_P] * 10 + [_I, _I, _I, _I, ctypes.c_double, _P, _I, _P, _P, _P, _P, _P],
    "cl3d_pwmlp_partials": [_I, _I, _I],
    "cl3d_pwmlp_pass_graphs": [_I],
    "cl3d_pwmlp_pass_graph_stats": [_P, _P],
    "cl3d_pwmlp_point_gemm_fwd": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _Z, _P],
    "cl3d_pwmlp_point_gemm_bwd_data": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P],
    "cl3d_pwmlp_point_gemm_bwd_weight": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P],
    "cl3d_pwmlp_point_gemm_bwd_fused": [_I, _I, _I, _I, _I],
    "cl3d_pwmlp_point_gemm_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _Z, _P],
    "cl3d_conv1x1_fwd": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P],
    "cl3d_conv1x1_bn_act_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _Z, _P],
    "cl3d_conv1x1_bwd_data": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P],
    "cl3d_conv1x1_bwd_weight": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P],
    "cl3d_pwmlp_point_gemm_fwd_pro": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _Z, _P],
    "cl3d_pwmlp_point_gemm_bwd_weight_pro": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P],
    "cl3d_conv1x1_rows_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P],
    "cl3d_conv1x1_rows_bwd_data": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P],
    "cl3d_conv1x1_rows_bwd_weight": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P],
    "cl3d_gemm_autotune": [_I],
    "cl3d_gemm_autotune_stats": [_P, _P],
    "cl3d_sgd_step": [_P, _P, _P, ctypes.c_longlong, _F, _F, _F, _F, _I, _I, _I, _P],
    "cl3d_pwmlp_split_weight": [_P, _I, _I, _P, _P, _P],
    "cl3d_pwmlp_merge_weight_grad": [_P, _P, _I, _I, _I, _P, _P],
    "cl3d_pwmlp_stats": [_P] * 6 + [_I] * 5 + [_F, _P, _P, _P, _P, _I, _P],
    "cl3d_pwmlp_finalize_stats": [_P, _I, _I, ctypes.c_double, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "cl3d_pwmlp_apply": [_P, _P, _P, _I, _I, _I, _P, _P],
    "cl3d_pwmlp_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _I, _P, _P, _P],
    "cl3d_pwmlp_bwd_rows": [_P, _I] + [_P] * 5 + [_F] + [_P] * 4 + [_I] * 5 + [_P, _P, _P, _P, _P, _I, _P],
    "cl3d_pwmlp_bwd_hits": [_P, _P, _I, _I, _I, _I, _P, _P],
    "cl3d_pwmlp_bn_backward_coeffs": [_P, _I, _I, ctypes.c_double] + [_P] * 11,
    "cl3d_pwmlp_bwd_hits_coeffs": [_P, _I, ctypes.c_double] + [_P] * 12 + [_I] * 4 + [_P, _P],
    "cl3d_pwmlp_bwd_support": [_P] * 10 + [_F, _P, _P] + [_I] * 5 + [_P, _P],
}


class Cl3dError(RuntimeError):
    pass


class PwmlpPass(ctypes.Structure):
    """cl3d_pwmlp_pass of include/cl3d.h (one C-ABI call per pass, csrc/pass.hip): field for field."""
    _fields_ = (
        [(n, _I) for n in ("B", "N", "M", "K", "C", "Co", "precision", "idx_ready", "csr_ready", "n_partials")]
        + [(n, _F) for n in ("radius", "eps", "momentum")]
        + [("reserved", _I)]
        + [(n, _P) for n in ("query_xyz", "support_xyz", "query_mask", "support_mask", "idx", "idx_mask", "inv_off",
                             "inv_slots", "bq_ws", "csr_ws", "gemm_ws", "gemm_ws_d", "gemm_ws_w")]
        + [(n, _Z) for n in ("bq_ws_bytes", "csr_ws_bytes", "gemm_ws_bytes", "gemm_ws_bytes_b")]
        + [(n, _P) for n in ("features", "W", "gamma", "beta", "running_mean", "running_var", "num_batches_tracked",
                             "ght", "wr", "wcat", "ystar", "sy", "vec", "out", "kstar", "partial", "sums", "partial_b",
                             "gout", "dz_cm", "dz_t", "qtab", "hit", "coef", "dwr", "dght", "dfeat", "dW", "ts_cm")])


class ReducePass(ctypes.Structure):
    """cl3d_reduce_pass of include/cl3d.h (PosPool / AdaptiveWeight / PseudoGrid, one C-ABI call per pass): field for field."""
    _fields_ = (
        [(n, _I) for n in ("B", "N", "M", "K", "C", "op", "normalize", "reduction", "pint", "constant", "idx_ready",
                           "csr_ready", "nparts")]
        + [(n, _F) for n in ("radius", "pfloat")]
        + [("reserved", _I)]
        + [(n, _P) for n in ("query_xyz", "support_xyz", "query_mask", "support_mask", "features", "p0", "p1", "idx",
                             "idx_mask", "inv_off", "inv_slots", "bq_ws", "csr_ws")]
        + [(n, _Z) for n in ("bq_ws_bytes", "csr_ws_bytes")]
        + [(n, _P) for n in ("ft", "out", "slotrec", "pairs", "gout", "gout_t", "dfeat", "dparam", "g0", "g1")]
        + [(n, _P) for n in ("gamma", "beta", "running_mean", "running_var", "num_batches_tracked", "act", "vec", "graw",
                             "coef", "bn_partial")]
        + [(n, _F) for n in ("eps", "momentum")]
        + [("bn_parts", _I), ("reserved2", _I)])


def _declare(handle):
    handle.cl3d_abi_version.restype = _I
    handle.cl3d_abi_version.argtypes = []
    handle.cl3d_d2_form.restype = _I
    handle.cl3d_d2_form.argtypes = []
    handle.cl3d_last_error_string.restype = ctypes.c_char_p
    handle.cl3d_last_error_string.argtypes = []
    handle.cl3d_workspace_bytes.restype = _Z
    handle.cl3d_workspace_bytes.argtypes = [_I] * 6
    for name, argtypes in SIGNATURES.items():
        fn = getattr(handle, name)
        fn.argtypes = argtypes
        fn.restype = _I
    for name in ("cl3d_pwmlp_train_forward", "cl3d_pwmlp_train_backward"):
        fn = getattr(handle, name)
        fn.argtypes = [ctypes.POINTER(PwmlpPass), _P]
        fn.restype = _I
    for name in ("cl3d_reduce_train_forward", "cl3d_reduce_train_backward"):
        fn = getattr(handle, name)
        fn.argtypes = [ctypes.POINTER(ReducePass), _P]
        fn.restype = _I


class _Traced:
    """Proxy around the library handle that brackets every stream-ordered entry point with a pair of HIP events on
    the stream the call is enqueued on (`trace()` below): per-entry-point GPU time of a step, measured live."""

    def __init__(self, handle, log):
        self._h, self._log, self._cache = handle, log, {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            raw = getattr(self._h, name)
            sig = SIGNATURES.get(name)
            if sig is None or sig[-1] is not _P:  # size queries etc.: no stream, nothing to time
                fn = raw
            else:
                log = self._log

                def fn(*args, _raw=raw, _name=name):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    st = torch.cuda.current_stream()  # == the stream argument every caller passes (stream_ptr)
                    e0.record(st)
                    rc = _raw(*args)
                    e1.record(st)
                    log.append((_name, e0, e1))
                    return rc
            self._cache[name] = fn
        return fn


_tracer = None


class trace:
    """`with _lib.trace() as t: step()` then `t.summary()` -> {entry point: (calls, total microseconds)}.
    Measurement only (bench.py's per-kernel table of the timed step); not for use under graph capture."""

    def __enter__(self):
        global _tracer
        self.log = []
        _tracer = _Traced(lib(), self.log)
        return self

    def __exit__(self, *exc):
        global _tracer
        _tracer = None
        return False

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1 in self.log:
            calls, us = out.get(name, (0, 0.0))
            out[name] = (calls + 1, us + e0.elapsed_time(e1) * 1e3)
        return out

    def per_run(self, runs):
        """{entry point: (calls per run, [microseconds of run 0, run 1, ...])} when the traced region was `runs`
        repetitions of the same step (every run makes the same calls in the same order)."""
        torch.cuda.synchronize()
        seen = {}
        for name, e0, e1 in self.log:
            seen.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
        out = {}
        for name, us in seen.items():
            per = max(1, len(us) // runs)
            out[name] = (per, [sum(us[r * per:(r + 1) * per]) for r in range(len(us) // per)])
        return out


# CL3D_ABI_VERSION of include/cl3d.h: the library must have been built from the same header (a stale .so with another
# argument layout would run with shifted pointers).  A constant, so that a copy of the package without the repository's
# include/ directory still imports; tests/test_abi.py holds it to the header.
ABI_VERSION = 5


def header_abi_version():
    """CL3D_ABI_VERSION as include/cl3d.h states it, or None where the header does not travel with the package."""
    import re
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "cl3d.h")
    if not os.path.exists(hdr):
        return None
    with open(hdr) as f:
        m = re.search(r"#define\s+CL3D_ABI_VERSION\s+(\d+)", f.read())
    return int(m.group(1)) if m else None


def lib():
    global _lib
    if _tracer is not None:
        return _tracer
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError(
                f"{_PATH} not found: the HIP engine is not built. Run `python -m closerlook3d_amd.build` "
                "(there is no CPU or PyTorch fallback for these ops).")
        handle = ctypes.CDLL(_PATH)
        _declare(handle)
        if handle.cl3d_abi_version() != ABI_VERSION:
            raise ImportError("libcl3d.so ABI version mismatch")
        if handle.cl3d_d2_form() != _D2_FORM:
            raise ImportError(f"{_PATH} was built with CL3D_D2_FORM={handle.cl3d_d2_form()}, expected {_D2_FORM}")
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise Cl3dError(f"cl3d error {rc}: {lib().cl3d_last_error_string().decode()}")


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _NoOp:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NOOP = _NoOp()


def on_device(device):
    """`with on_device(t.device):` -- torch.cuda.device(...) only when it would actually switch devices (the
    context manager costs ~5 us of host time per use, and a step makes dozens of engine calls)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NOOP
    return torch.cuda.device(device)

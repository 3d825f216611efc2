"""`pt_custom_ops._ext` surface on top of libcl3d (reference: _ext_src/src/bindings.cpp:8-14).

Same five function names, argument order, dtypes, shapes and error behaviour as the reference's
pybind module: float32/int32, contiguous, on the GPU; anything else raises RuntimeError with the
reference's wording ("... must be a contiguous tensor", "CPU not supported", ...), see
_ext_src/include/utils.h:10-30.  Outputs are allocated here (torch is the allocator / stream
provider, nothing more) and filled by the HIP kernels through the C ABI.
"""
import torch

from . import _lib


def _check(name, t, dtype):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {'a float' if dtype == torch.float32 else 'an int'} tensor")


def _check_dev(ref, **others):
    if not ref.is_cuda:
        raise RuntimeError("CPU not supported")
    for name, t in others.items():
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
        if t.device != ref.device:
            raise RuntimeError(f"{name} is on {t.device}, expected {ref.device}")


def _p(t):
    return t.data_ptr()


def group_points(points, idx):
    _check("points", points, torch.float32)
    _check("idx", idx, torch.int32)
    _check_dev(points, idx=idx)
    B, C, N = points.shape
    _, M, K = idx.shape
    out = torch.empty((B, C, M, K), dtype=torch.float32, device=points.device)
    with _lib.on_device(points.device):
        _lib.check(_lib.lib().cl3d_group_points(_p(points), _p(idx), B, C, N, M, K, _p(out),
                                                _lib.stream_ptr(points.device)))
    return out


def group_points_grad(grad_out, idx, n):
    _check("grad_out", grad_out, torch.float32)
    _check("idx", idx, torch.int32)
    _check_dev(grad_out, idx=idx)
    B, C, M, K = grad_out.shape
    out = torch.empty((B, C, int(n)), dtype=torch.float32, device=grad_out.device)
    with _lib.on_device(grad_out.device):
        _lib.check(_lib.lib().cl3d_group_points_grad(_p(grad_out), _p(idx), B, C, int(n), M, K, _p(out),
                                                     None, 0, _lib.stream_ptr(grad_out.device)))
    return out


BQ_PATHS = {"auto": 0, "tile": 1, "cells": 2, "exhaustive": 3}


def masked_ordered_ball_query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample, path=0):
    """The reference's signature; `path` (engine extension, default 0 = the library's choice) names one implementation
    (BQ_PATHS): every path returns the same bits (tests/test_bq_paths_gpu.py), pt_utils picks by measurement."""
    _check("query_xyz", query_xyz, torch.float32)
    _check("support_xyz", support_xyz, torch.float32)
    _check("query_mask", query_mask, torch.int32)
    _check("support_mask", support_mask, torch.int32)
    _check_dev(query_xyz, support_xyz=support_xyz, query_mask=query_mask, support_mask=support_mask)
    B, M, _ = query_xyz.shape
    N = support_xyz.shape[1]
    idx = torch.empty((B, M, int(nsample)), dtype=torch.int32, device=query_xyz.device)
    idx_mask = torch.empty_like(idx)
    lib = _lib.lib()
    ws_bytes = lib.cl3d_workspace_bytes(1, B, N, M, int(nsample), 0)  # CL3D_OP_BALL_QUERY
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=query_xyz.device) if ws_bytes else None
    with _lib.on_device(query_xyz.device):
        _lib.check(lib.cl3d_masked_ordered_ball_query_path(
            int(path), _p(query_xyz), _p(support_xyz), _p(query_mask), _p(support_mask), B, M, N, float(radius),
            int(nsample), _p(idx), _p(idx_mask), _p(ws) if ws is not None else None, ws_bytes,
            _lib.stream_ptr(query_xyz.device)))
    return [idx, idx_mask]


def masked_grid_subsampling(points, mask, nsamples, sampleDl):
    _check("points", points, torch.float32)
    _check("mask", mask, torch.int32)
    _check_dev(points, mask=mask)
    B, N, _ = points.shape
    sub = torch.empty((B, int(nsamples), 3), dtype=torch.float32, device=points.device)
    sub_mask = torch.empty((B, int(nsamples)), dtype=torch.int32, device=points.device)
    lib = _lib.lib()
    ws_bytes = lib.cl3d_workspace_bytes(4, B, N, 0, 0, 0)  # CL3D_OP_GRID_SUBSAMPLING (non-zero for N > 16384)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=points.device) if ws_bytes else None
    with _lib.on_device(points.device):
        _lib.check(lib.cl3d_masked_grid_subsampling(
            _p(points), _p(mask), B, N, int(nsamples), float(sampleDl), _p(sub), _p(sub_mask),
            _p(ws) if ws is not None else None, ws_bytes, _lib.stream_ptr(points.device)))
    return [sub, sub_mask]


def masked_nearest_query(query_xyz, support_xyz, query_mask, support_mask):
    _check("query_xyz", query_xyz, torch.float32)
    _check("support_xyz", support_xyz, torch.float32)
    _check("query_mask", query_mask, torch.int32)
    _check("support_mask", support_mask, torch.int32)
    _check_dev(query_xyz, support_xyz=support_xyz, query_mask=query_mask, support_mask=support_mask)
    B, M, _ = query_xyz.shape
    N = support_xyz.shape[1]
    idx = torch.empty((B, M, 1), dtype=torch.int32, device=query_xyz.device)
    idx_mask = torch.empty_like(idx)
    with _lib.on_device(query_xyz.device):
        _lib.check(_lib.lib().cl3d_masked_nearest_query(
            _p(query_xyz), _p(support_xyz), _p(query_mask), _p(support_mask), B, M, N, _p(idx),
            _p(idx_mask), _lib.stream_ptr(query_xyz.device)))
    return [idx, idx_mask]


# ---- beyond the legacy five: fused grouping used by pt_utils.MaskedQueryAndGroup ----------------
def group_xyz_features(query_xyz, support_xyz, features, idx, radius, normalize_xyz):
    """rel [B,3,M,K] (and grouped [B,C,M,K] if features is not None) in one call."""
    _check("query_xyz", query_xyz, torch.float32)
    _check("support_xyz", support_xyz, torch.float32)
    _check("idx", idx, torch.int32)
    _check_dev(query_xyz, support_xyz=support_xyz, idx=idx)
    B, M, K = idx.shape
    N = support_xyz.shape[1]
    rel = torch.empty((B, 3, M, K), dtype=torch.float32, device=idx.device)
    grouped, C, fptr, gptr = None, 0, None, None
    if features is not None:
        _check("features", features, torch.float32)
        _check_dev(query_xyz, features=features)
        C = features.shape[1]
        grouped = torch.empty((B, C, M, K), dtype=torch.float32, device=idx.device)
        fptr, gptr = _p(features), _p(grouped)
    with _lib.on_device(idx.device):
        _lib.check(_lib.lib().cl3d_group_xyz_features(
            _p(query_xyz), _p(support_xyz), fptr, _p(idx), B, C, N, M, K, float(radius),
            1 if normalize_xyz else 0, _p(rel), gptr, _lib.stream_ptr(idx.device)))
    return rel, grouped

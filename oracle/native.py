"""ctypes/numpy binding of oracle/cl3d_oracle.c (test infrastructure, see oracle/__init__.py).

Signatures mirror the reference's pybind functions (bindings.cpp:8-14) on numpy arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def d2_form():
    """CL3D_D2_FORM of this process (0 = the canon; see cl3d_oracle.c and closerlook3d_amd/build.py)."""
    return int(os.environ.get("CL3D_D2_FORM", "0") or 0)


def build(force=False, form=None):
    form = d2_form() if form is None else form
    name = "libcl3d_oracle.so" if form == 0 else f"libcl3d_oracle_d2form{form}.so"
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "cl3d_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", name], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def set_threads(n=0):
    """OpenMP threads of the native-op loops (batch x query / batch x channel); 0 keeps the runtime default.
    Returns the count in effect.  Results do not depend on it."""
    fn = lib().oracle_set_threads
    fn.restype = ctypes.c_int
    return int(fn(int(n)))


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def masked_ordered_ball_query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample):
    q, qp = _f32(query_xyz)
    s, sp = _f32(support_xyz)
    qm, qmp = _i32(query_mask)
    sm, smp = _i32(support_mask)
    B, M, _ = q.shape
    N = s.shape[1]
    idx = np.zeros((B, M, nsample), np.int32)
    msk = np.zeros((B, M, nsample), np.int32)
    rc = lib().oracle_masked_ordered_ball_query(
        qp, sp, qmp, smp, B, M, N, ctypes.c_float(radius), int(nsample),
        idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), msk.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    assert rc == 0
    return idx, msk


def group_points(points, idx):
    p, pp = _f32(points)
    i, ip = _i32(idx)
    B, C, N = p.shape
    _, M, K = i.shape
    out = np.zeros((B, C, M, K), np.float32)
    rc = lib().oracle_group_points(pp, ip, B, C, N, M, K, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rc == 0
    return out


def group_points_grad(grad_out, idx, n):
    g, gp = _f32(grad_out)
    i, ip = _i32(idx)
    B, C, M, K = g.shape
    out = np.zeros((B, C, n), np.float32)
    rc = lib().oracle_group_points_grad(gp, ip, B, C, int(n), M, K,
                                        out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    assert rc == 0
    return out


def masked_grid_subsampling(points, mask, nsamples, sampleDl):
    p, pp = _f32(points)
    m, mp = _i32(mask)
    B, N, _ = p.shape
    sub = np.zeros((B, nsamples, 3), np.float32)
    smask = np.zeros((B, nsamples), np.int32)
    rc = lib().oracle_masked_grid_subsampling(
        pp, mp, B, N, int(nsamples), ctypes.c_float(sampleDl),
        sub.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), smask.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    assert rc == 0
    return sub, smask


def masked_nearest_query(query_xyz, support_xyz, query_mask, support_mask):
    q, qp = _f32(query_xyz)
    s, sp = _f32(support_xyz)
    qm, qmp = _i32(query_mask)
    sm, smp = _i32(support_mask)
    B, M, _ = q.shape
    N = s.shape[1]
    idx = np.zeros((B, M, 1), np.int32)
    msk = np.zeros((B, M, 1), np.int32)
    rc = lib().oracle_masked_nearest_query(
        qp, sp, qmp, smp, B, M, N,
        idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), msk.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    assert rc == 0
    return idx, msk


def dataset_grid_subsampling(points, features=None, labels=None, sampleDl=0.1):
    """Dataset-side grid subsampling (datasets/data_utils.py:12-30 -> grid_subsampling.cpp), voxels in ascending
    (iz, iy, ix) order.  Returns (sub_points [m,3], sub_features [m,fdim] | None, sub_labels [m,ldim] | None)."""
    p, pp = _f32(points)
    n = p.shape[0]
    fdim = 0 if features is None else np.asarray(features).shape[1]
    ldim = 0 if labels is None else np.asarray(labels).shape[1]
    f, fp = _f32(features if features is not None else np.zeros((1, 1)))
    l, lp = _i32(labels if labels is not None else np.zeros((1, 1)))
    sp = np.zeros((n, 3), np.float32)
    sf = np.zeros((n, max(fdim, 1)), np.float32)
    sl = np.zeros((n, max(ldim, 1)), np.int32)
    fn = lib().oracle_dataset_grid_subsampling
    fn.restype = ctypes.c_int
    m = fn(pp, fp, lp, n, fdim, ldim, ctypes.c_float(sampleDl), sp.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
           sf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), sl.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return sp[:m], (sf[:m, :fdim] if fdim else None), (sl[:m, :ldim] if ldim else None)

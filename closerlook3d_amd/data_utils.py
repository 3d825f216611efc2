"""Dataset-side helpers on the GPU.  `grid_subsampling` has the signature of the reference's
datasets/data_utils.py:12-30 (which calls the host C++ of ops/cpp_wrappers/cpp_subsampling): numpy in -> numpy
out, or CUDA tensors in -> CUDA tensors out."""
import numpy as np
import torch

from . import _lib


def grid_subsampling(points, features=None, labels=None, sampleDl=0.1, verbose=0):
    """Voxel-grid subsampling of one cloud: barycentre of the points of every voxel of size `sampleDl`, the mean of
    their features and their most frequent label (per label column).

    Returns what the reference returns: `sub_points`, then `sub_features` and/or `sub_labels` when given.  Voxels
    are emitted in ascending (iz, iy, ix) order and label ties go to the smallest label (both are unspecified in
    the reference, whose result follows an unordered_map walk); sums run in original point order, so coordinates
    and features are bit-identical to the reference's."""
    as_numpy = not torch.is_tensor(points)
    dev = torch.device("cuda", torch.cuda.current_device()) if as_numpy else points.device
    if dev.type != "cuda":
        raise RuntimeError("CPU not supported")

    def to_dev(a, dtype):
        if a is None:
            return None
        t = torch.from_numpy(np.ascontiguousarray(a)) if not torch.is_tensor(a) else a
        return t.to(device=dev, dtype=dtype).contiguous()

    p = to_dev(points, torch.float32)
    f = to_dev(features, torch.float32)
    lb = to_dev(labels, torch.int32)
    if p.dim() != 2 or p.shape[1] != 3:
        raise RuntimeError("points must be [N,3]")
    N = p.shape[0]
    fdim = 0 if f is None else f.reshape(N, -1).shape[1]
    ldim = 0 if lb is None else lb.reshape(N, -1).shape[1]
    sp = torch.empty((N, 3), dtype=torch.float32, device=dev)
    sf = torch.empty((N, fdim), dtype=torch.float32, device=dev) if fdim else None
    sl = torch.empty((N, ldim), dtype=torch.int32, device=dev) if ldim else None
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    lib = _lib.lib()
    ws_bytes = lib.cl3d_workspace_bytes(12, 1, N, 0, 0, 0)  # CL3D_OP_DATASET_GRID
    ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=dev)
    ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    with _lib.on_device(dev):
        _lib.check(lib.cl3d_dataset_grid_subsampling(ptr(p), ptr(f), ptr(lb), N, fdim, ldim, float(sampleDl), ptr(sp),
                                                     ptr(sf), ptr(sl), ptr(count), ptr(ws), ws_bytes,
                                                     _lib.stream_ptr(dev)))
    m = int(count.item())  # the one host round trip: the output size is data-dependent
    out = [sp[:m]]
    if sf is not None:
        out.append(sf[:m])
    if sl is not None:
        out.append(sl[:m])
    if as_numpy:
        out = [t.cpu().numpy() for t in out]
    return out[0] if len(out) == 1 else tuple(out)

"""One C-ABI call per pass of a LocalAggregation operator in training mode: the autograd nodes over
cl3d_pwmlp_train_forward / _backward (PointWiseMLP) and cl3d_reduce_train_forward / _backward (PosPool / AdaptiveWeight /
PseudoGrid, round 6) of csrc/pass.hip.  fused.pointwise_mlp() / pospool() / adaptive_weight() / pseudo_grid() take this path
for a stand-alone operator launched eagerly (outside HIP-graph capture); everything else goes kernel by kernel through
fused._PointwiseMLP / fused._FusedReduce."""
import ctypes

import torch
from torch.autograd import Function

from . import _lib


def _stream(t):
    return _lib.stream_ptr(t.device)


class _PointwiseMLPPass(Function):
    """The whole PointWiseMLP operator in training mode -- ball query, CSR inverse, per-point product, statistics pass,
    BatchNorm, activation; and its whole backward -- as ONE C-ABI call per direction (csrc/pass.hip,
    cl3d_pwmlp_train_forward / _backward): the library enqueues every kernel itself, the geometry work and the weight
    gradient on its own side streams.  This is the eager caller's path (the reference's unchanged training loop makes one
    `_ext` call per autograd node, pt_utils.py:16-61): launched kernel by kernel from Python the host sets the pace
    (0.41 ms per step at the metric shape, 0.59 ms with the forks made from Python), a captured step keeps the
    Python-side schedule (`pointwise_mlp` below).  Same kernels, same arithmetic, same bits as that path."""

    @staticmethod
    def forward(ctx, features, W, gamma, beta, running_mean, running_var, num_batches_tracked, query_xyz, support_xyz,
                query_mask, support_mask, radius, nsample, momentum, eps, precision, need_grad):
        B, C, N = features.shape
        M, K, Co = query_xyz.shape[1], int(nsample), W.shape[0]
        dev = features.device
        lib = _lib.lib()
        p = _lib.PwmlpPass()
        p.B, p.N, p.M, p.K, p.C, p.Co, p.precision = B, N, M, K, C, Co, precision
        p.radius, p.eps, p.momentum = float(radius), float(eps), float(momentum)
        p.idx_ready = p.csr_ready = 0
        inputs = (features, W, gamma, beta, running_mean, running_var, num_batches_tracked, query_xyz, support_xyz,
                  query_mask, support_mask)  # kept alive on the node: the argument block points into them
        for name, t in zip(("features", "W", "gamma", "beta", "running_mean", "running_var", "num_batches_tracked",
                            "query_xyz", "support_xyz", "query_mask", "support_mask"), inputs):
            setattr(p, name, t.data_ptr())
        p.bq_ws_bytes = lib.cl3d_workspace_bytes(1, B, N, M, K, 0)  # CL3D_OP_BALL_QUERY
        p.csr_ws_bytes = lib.cl3d_workspace_bytes(11, B, N, M * K, 1, 0) if need_grad else 0  # CL3D_OP_INVERSE_INDEX
        p.gemm_ws_bytes = lib.cl3d_workspace_bytes(14, B, N, Co, 0, C)  # CL3D_OP_POINT_GEMM
        p.n_partials = lib.cl3d_pwmlp_partials(B, M, Co)
        # everything the pass leaves behind for its backward, and its scratch, is ONE allocation (the host sets the pace
        # of an eager step: two dozen torch.empty calls cost ~40 us of it); `out` is a tensor of its own
        arena = _Arena(p)
        arena.add("idx", 4 * B * M * K)
        arena.add("idx_mask", 4 * B * M * K)
        arena.add("bq_ws", p.bq_ws_bytes)
        if need_grad:
            arena.add("inv_off", 4 * B * (N + 1))
            arena.add("inv_slots", 4 * B * M * K)
            arena.add("csr_ws", p.csr_ws_bytes)
        arena.add("gemm_ws", p.gemm_ws_bytes)
        arena.add("ght", 4 * B * N * 2 * Co)
        arena.add("wr", 4 * Co * 3)
        arena.add("wcat", 4 * 2 * Co * C)
        arena.add("ystar", 4 * B * M * Co)
        arena.add("sy", 4 * B * M * Co)
        arena.add("kstar", B * M * Co)
        arena.add("partial", 8 * p.n_partials * Co * 8)
        arena.add("vec", 4 * 4 * Co)
        arena.add("sums", 8 * Co * 6)
        kept = arena.allocate(dev)
        out = torch.empty((B, Co, M), dtype=torch.float32, device=dev)
        p.out = out.data_ptr()
        with _lib.on_device(dev):
            _lib.check(lib.cl3d_pwmlp_train_forward(ctypes.byref(p), _stream(features)))
        ctx.block, ctx.keep, ctx.need = p, [inputs, kept], need_grad  # (not `out`: the node must not own its own output)
        return out

    @staticmethod
    def backward(ctx, gout):
        p = ctx.block
        B, N, M, C, Co = p.B, p.N, p.M, p.C, p.Co
        dev = gout.device
        lib = _lib.lib()
        gout = gout.contiguous()
        p.gout = gout.data_ptr()
        p.gemm_ws_bytes_b = lib.cl3d_workspace_bytes(14, B, N, Co, 0, C)
        arena = _Arena(p)
        arena.add("dz_cm", 4 * B * Co * M)
        arena.add("ts_cm", 4 * B * Co * M)
        arena.add("dz_t", 4 * B * M * Co)
        arena.add("qtab", 16 * B * M)
        arena.add("partial_b", 8 * p.n_partials * Co * 8)
        arena.add("hit", 4 * B * Co * N)
        arena.add("dwr", 4 * Co * 3)
        arena.add("dght", 4 * B * N * 2 * Co)
        arena.add("gemm_ws_d", p.gemm_ws_bytes_b)
        arena.add("gemm_ws_w", p.gemm_ws_bytes_b)
        scratch = arena.allocate(dev)
        coef = torch.empty((5, Co), dtype=torch.float32, device=dev)
        p.coef = coef.data_ptr()
        dfeat = torch.empty((B, C, N), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dW = torch.empty((Co, 3 + 2 * C), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        p.dfeat = dfeat.data_ptr() if dfeat is not None else None
        p.dW = dW.data_ptr() if dW is not None else None
        with _lib.on_device(dev):
            _lib.check(lib.cl3d_pwmlp_train_backward(ctypes.byref(p), _stream(gout)))
        del scratch  # (everything that used it is queued on -- or joined into -- the stream whose allocations reuse it)
        # (ctx.keep stays: autograd drops the node -- and with it the forward's buffers -- unless the caller retains the
        # graph, in which case a second backward pass reads them again)
        return (dfeat, dW, coef[3], coef[4]) + (None,) * 13


class _ReducePass(Function):
    """A PosPool / AdaptiveWeight / PseudoGrid operator up to its output transform -- ball query, CSR inverse, layout
    change, the fused reduction; and its whole backward -- as ONE C-ABI call per direction (csrc/pass.hip,
    cl3d_reduce_train_forward / _backward; round 6): the eager caller's path for these three operators,
    as _PointwiseMLPPass is for the fourth.  Same kernels, same arithmetic, same bits as fused._FusedReduce."""

    @staticmethod
    def forward(ctx, features, p0, p1, op, query_xyz, support_xyz, query_mask, support_mask, radius, nsample, normalize,
                reduction, pint, pfloat, constant, gamma=None, beta=None, running_mean=None, running_var=None,
                num_batches_tracked=None, momentum=0.1, eps=1e-5):
        """gamma ... eps (optional): the operator's BatchNorm1d + ReLU output transform rides in the same two calls; the
        node then returns the ACTIVATED output and its backward also yields d gamma, d beta."""
        B, C, N = features.shape
        M, K = query_xyz.shape[1], int(nsample)
        dev = features.device
        lib = _lib.lib()
        p = _lib.ReducePass()
        p.B, p.N, p.M, p.K, p.C, p.op = B, N, M, K, C, int(op)
        p.normalize, p.reduction, p.pint, p.constant = int(normalize), int(reduction), int(pint), int(constant)
        p.radius, p.pfloat = float(radius), float(pfloat)
        p.idx_ready = p.csr_ready = 0
        inputs = (features, p0, p1, query_xyz, support_xyz, query_mask, support_mask)  # kept alive: the block points into them
        for name, t in zip(("features", "p0", "p1", "query_xyz", "support_xyz", "query_mask", "support_mask"), inputs):
            setattr(p, name, t.data_ptr() if t is not None else None)
        p.bq_ws_bytes = lib.cl3d_workspace_bytes(1, B, N, M, K, 0)            # CL3D_OP_BALL_QUERY
        p.csr_ws_bytes = lib.cl3d_workspace_bytes(11, B, N, M * K, 1, 0)      # CL3D_OP_INVERSE_INDEX
        p.nparts = lib.cl3d_fused_param_partials(int(op), B, N, C)
        arena = _Arena(p)
        arena.add("idx", 4 * B * M * K)
        arena.add("idx_mask", 4 * B * M * K)
        arena.add("bq_ws", p.bq_ws_bytes)
        arena.add("inv_off", 4 * B * (N + 1))
        arena.add("inv_slots", 4 * B * M * K)
        arena.add("csr_ws", p.csr_ws_bytes)
        arena.add("ft", 4 * B * N * C)
        arena.add("slotrec", 16 * B * M * K)
        sparse = int(op) == 3 and C % 4 == 0 and not constant  # PseudoGrid's (kernel point, influence) pairs
        if sparse:
            arena.add("pairs", 32 * B * M * K)
        with_bn = gamma is not None
        if with_bn:
            p.bn_parts = lib.cl3d_bn_partials(B, C, M)
            p.eps, p.momentum = float(eps), float(momentum)
            arena.add("out", 4 * B * C * M)                       # the raw result: kept for the BatchNorm backward
            arena.add("vec", 4 * 4 * C)
            arena.add("bn_partial", 8 * 2 * p.bn_parts * C * 2)
            bn_inputs = (gamma, beta, running_mean, running_var, num_batches_tracked)
            for name, t in zip(("gamma", "beta", "running_mean", "running_var", "num_batches_tracked"), bn_inputs):
                setattr(p, name, t.data_ptr() if t is not None else None)
            inputs = inputs + bn_inputs
        kept = arena.allocate(dev)
        if not sparse:
            p.pairs = None
        out = torch.empty((B, C, M), dtype=torch.float32, device=dev)
        if with_bn:
            p.act = out.data_ptr()
        else:
            p.out = out.data_ptr()
        with _lib.on_device(dev):
            _lib.check(lib.cl3d_reduce_train_forward(ctypes.byref(p), _stream(features)))
        ctx.block, ctx.keep, ctx.with_bn = p, [inputs, kept], with_bn
        if with_bn:
            ctx.save_for_backward(out)  # (the activated output gates the ReLU in the backward pass)
        return out

    @staticmethod
    def backward(ctx, gout):
        p = ctx.block
        B, N, M, C, op = p.B, p.N, p.M, p.C, p.op
        dev = gout.device
        lib = _lib.lib()
        gout = gout.contiguous()
        p.gout = gout.data_ptr()
        npar = {2: 4, 3: 16}.get(op, 0)
        arena = _Arena(p)
        arena.add("gout_t", 4 * B * M * C)
        arena.add("dparam", 4 * p.nparts * C * npar)
        coef = None
        if ctx.with_bn:
            (act,) = ctx.saved_tensors
            p.act = act.data_ptr()
            arena.add("graw", 4 * B * C * M)
            coef = torch.empty((5, C), dtype=torch.float32, device=dev)
            p.coef = coef.data_ptr()
        scratch = arena.allocate(dev)
        dfeat = torch.empty((B, C, N), dtype=torch.float32, device=dev)
        p.dfeat = dfeat.data_ptr()
        g0 = g1 = None
        if op == 2:
            g0 = torch.empty((C // p.pint, 3), dtype=torch.float32, device=dev)
            g1 = torch.empty((C // p.pint,), dtype=torch.float32, device=dev)
        elif op == 3:
            g1 = torch.empty((p.pint, C), dtype=torch.float32, device=dev)
        p.g0 = g0.data_ptr() if g0 is not None else None
        p.g1 = g1.data_ptr() if g1 is not None else None
        with _lib.on_device(dev):
            _lib.check(lib.cl3d_reduce_train_backward(ctypes.byref(p), _stream(gout)))
        del scratch
        if ctx.with_bn:
            return (dfeat, g0, g1) + (None,) * 12 + (coef[3], coef[4]) + (None,) * 5
        return (dfeat, g0, g1) + (None,) * 12 + (None,) * 7


class _Arena:
    """Named sub-buffers of ONE uint8 allocation, 256-byte aligned, their addresses written into the fields of an
    argument block (a zero-size buffer keeps a valid, unused address)."""

    def __init__(self, block):
        self.block, self.items, self.size = block, [], 0

    def add(self, name, nbytes):
        self.items.append((name, self.size))
        self.size += (max(int(nbytes), 1) + 255) & ~255

    def allocate(self, device):
        buf = torch.empty((self.size,), dtype=torch.uint8, device=device)
        base = buf.data_ptr()
        for name, off in self.items:
            setattr(self.block, name, base + off)
        return buf

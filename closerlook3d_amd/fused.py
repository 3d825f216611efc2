"""Fused local-aggregation operators: Python side of csrc/fused_reduce.hip, fused_pwmlp.hip, csr.hip.

The reference's operators materialise the grouped neighbourhood tensor [B,C,M,K] and run element-wise
PyTorch ops over it (models/local_aggregation_operators.py).  These entry points compute the same
`[B, C_out, M]` result (up to the output transform, which stays ordinary PyTorch) from point-major
feature rows with hand-written HIP kernels, forward and backward, without that tensor.

`use_fused(impl, kind, module)` decides per operator instance; configurations the fused kernels do not
cover (non-shipped variants such as PosPool with max reduction or a two-layer AdaptiveWeight MLP) run the
'grouped' dataflow instead -- still on the engine's native ops -- and `impl='fused'` raises for them.
"""
import contextlib
import os

import torch
from torch.autograd import Function

from . import _lib
from . import pt_utils
from .pt_utils import _ball_query, wait_ready

OP_POSPOOL_XYZ, OP_POSPOOL_SINCOS, OP_ADAPTIVE, OP_PSEUDOGRID = 0, 1, 2, 3
_RED = {'sum': 0, 'avg': 1, 'mean': 1}


_OP_ID = {'pospool': 7, 'adaptive_weight': 8, 'pseudo_grid': 9, 'pointwisemlp': 10, 'max_pool': 13}  # CL3D_OP_*


def kernels_cover(kind, nsample, channels):
    """The fused kernels keep a block's slot tile in LDS, which grows with nsample: ask the library."""
    return bool(_lib.lib().cl3d_fused_supported(_OP_ID[kind], int(nsample), int(channels)))


def _bn_ok(bn):
    # momentum=None means a cumulative moving average in PyTorch; the kernels implement the exponential rule only
    return bn.momentum is not None


def _supported(kind, m):
    if kind == 'pospool':
        ok = m.position_embedding in ('xyz', 'sin_cos') and m.reduction in _RED
        return ok and kernels_cover(kind, m.nsample, m.in_channels)
    if kind == 'adaptive_weight':
        ok = m.weight_type == 'dp' and m.num_mlps == 1 and m.reduction in _RED
        return ok and kernels_cover(kind, m.nsample, m.in_channels)
    if kind == 'pointwisemlp':
        ok = m.feature_type == 'dp_fi_df' and m.num_mlps == 1 and m.reduction == 'max' and m.nsample <= 255
        return ok and _bn_ok(m.mlps.conv0[1]) and kernels_cover(kind, m.nsample, m.out_channels)
    if kind == 'pseudo_grid':
        ok = m.KP_influence in ('linear', 'constant') and m.num_kernel_points <= 16 and m.convolution_mode == 'sum'
        return ok and kernels_cover(kind, m.nsample, m.in_channels)
    return False


def use_fused(impl, kind, module):
    if impl == 'grouped':
        return False
    ok = _supported(kind, module)
    if kind == 'pointwisemlp' and ok and not module.training and torch.is_grad_enabled():
        ok = False  # backward through frozen BatchNorm statistics: only the grouped path implements it
    if impl == 'fused' and not ok:
        raise NotImplementedError(f"fused path does not cover this '{kind}' configuration")
    return ok


def _p(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    return _lib.stream_ptr(t.device)


FORK_MIN_POINTS = 0  # A/B: contractions over fewer points (B * N) than this run their two gradient products in line
_DEFER = [False]      # deferred_weight_gradients(): weight gradients stay on the side stream until join_weight_gradients()
_DEFERRED = []        # (event, parameter, d W, side stream) of the weight gradients not joined yet


def _weight_leaf(W):
    """The nn.Parameter a contraction's weight argument is a plain reshape of (conv.weight.view(Co, -1)), or None: the
    tensor a deferred weight gradient is handed to."""
    base = W if W._base is None else W._base
    if base.is_leaf and base.requires_grad and base.numel() == W.numel() and W.is_contiguous() and base.is_contiguous():
        return base
    return None


def _fork_join(device, side_fn, main_fn, points=None, defer=None):
    """Two independent pieces of a backward pass (the weight gradient and the data gradient of one contraction) side by
    side: `side_fn` on a side HIP stream, `main_fn` on the caller's, joined before returning.  Active only while a
    DECLARED whole-step HIP graph is captured (whole_step_capture(): forward and backward in one capture), where the
    fork becomes two parallel branches of the graph; everywhere else -- eager launches, a capture that holds a backward
    pass alone (torch.cuda.make_graphed_callables) -- the two run one after the other on the caller's stream.  A forked
    pair inside a graph that holds ONLY a backward pass gave replay-varying gradients in rounds 3-5; round 6 traced it to
    a THIRD concurrent branch in that graph: autograd runs every AccumulateGrad node on the stream that was current when
    the parameter was first used -- the warm-up's, when the warm-up has a stream of its own, which is how
    torch.cuda.make_graphed_callables and the usual torch.cuda.graph recipe work.  With warm-up and capture on ONE stream
    (closerlook3d_amd.step_stream) the same forks are exact (profiles/r06/two_graph_repeat_check.txt).  A capture
    this module knows nothing about may still have that third branch, so an undeclared capture stays single-stream here.
    `side_fn` is the LONGER piece (callers pass the weight gradient: product + slice reduce) and is enqueued FIRST: the
    HIP runtime lays a graph out depth-first along each node's first-captured dependent (that one inherits the node's
    queue, the next one goes to the other queue; scripts/micro/graph_queues.hip), so the piece captured first -- and the
    join node behind it -- stay on the predecessor's queue, and it is the SHORT piece that pays the two cross-queue
    hand-overs, which it can afford (measured the other way round in round 5: the config-2 backbone 8.19 -> 8.39 ms).
    Outputs are allocated by the caller BEFORE the fork (on the caller's stream); whatever side_fn allocates is scratch
    that lives and dies on the side stream.

    `defer` = (parameter, d W, tensors side_fn reads), under deferred_weight_gradients() only: the join is NOT taken here.
    Nothing of the backward pass reads a weight gradient, so the side stream keeps running it beside the layers that
    follow and join_weight_gradients() -- called by the step between backward() and the optimizer -- joins the side stream
    once and hands every d W to its parameter's .grad.  Returns True when the gradient was deferred: the caller then
    returns None for the weight (autograd has nothing to accumulate).  What side_fn reads is marked as in use on the side
    stream (record_stream): the caller's stream frees those tensors long before the side stream has read them."""
    if not (device.type == 'cuda' and pt_utils.async_index() and _forks_allowed()) or (
            points is not None and points < FORK_MIN_POINTS):
        side_fn()
        main_fn()
        return False
    main, side = torch.cuda.current_stream(device), pt_utils.index_stream(device, 2)
    if _DEFER[0] and defer is not None and defer[0] is not None and defer[1] is not None:
        # deferred: the caller's piece is captured FIRST -- it is then the predecessor's first dependent and stays on its
        # queue (the rule above), so the chain the backward pass waits for never changes queue at a fork; the weight
        # gradient, which nothing waits for until the step's join, takes the hand-over
        at_fork = torch.cuda.Event()
        at_fork.record(main)
        main_fn()
        side.wait_event(at_fork)
        with torch.cuda.stream(side):
            side_fn()
            ev = torch.cuda.Event()
            ev.record(side)
        for t in defer[2]:
            if t is not None:
                t.record_stream(side)
        _DEFERRED.append((ev, defer[0], defer[1], side))
        return True
    side.wait_stream(main)
    with torch.cuda.stream(side):
        side_fn()
        ev = torch.cuda.Event()
        ev.record(side)
    main_fn()
    main.wait_event(ev)
    return False


@contextlib.contextmanager
def deferred_weight_gradients(on=True):
    """Inside a declared whole-step capture (whole_step_capture()): the weight gradients of the engine's contractions (the
    1x1 convolutions, the PointWiseMLP's per-point product) are not joined where they are formed.  The step must call
    join_weight_gradients() after backward() and before anything reads a parameter's .grad (the optimizer, a gradient
    exchange); a capture that ends without it fails loudly (hipErrorStreamCaptureUnjoined, reported by name here).  The
    deferred gradients reach .grad directly -- parameter hooks do not see them -- and only for weights that are plain
    reshapes of a leaf parameter; every other weight gradient is joined in place as before."""
    old = _DEFER[0]
    _DEFER[0] = bool(on)
    try:
        yield
        if _DEFERRED:
            n = len(_DEFERRED)
            join_weight_gradients()
            raise RuntimeError(f"deferred_weight_gradients(): {n} weight gradient(s) were still on the side stream when the "
                               "context ended -- call closerlook3d_amd.join_weight_gradients() after backward()")
    finally:
        _DEFER[0] = old
        del _DEFERRED[:]


def join_weight_gradients():
    """The caller's stream picks up the side stream that ran the deferred weight gradients (one wait per device: the side
    stream is in order) and every d W is handed to its parameter: .grad = d W where it was None, .grad += d W (on the
    caller's stream, behind the wait) where a gradient is accumulated -- flat gradient buffers (dp.FlatGradients) included.
    Returns how many gradients it delivered."""
    if not _DEFERRED:
        return 0
    last = {}
    for ev, _, dW, _ in _DEFERRED:
        last[dW.device] = ev
    for dev, ev in last.items():
        torch.cuda.current_stream(dev).wait_event(ev)
    with torch.no_grad():
        into, what = [], []
        for _, param, dW, _ in _DEFERRED:
            g = dW.view_as(param)
            if param.grad is None:
                param.grad = g
            else:
                into.append(param.grad)
                what.append(g)
        if into:  # accumulated gradients (flat buffers): one multi-tensor add for all of them, behind the wait
            torch._foreach_add_(into, what)
    n = len(_DEFERRED)
    del _DEFERRED[:]
    return n


def _gemm_scratch(op, B, N, Co, C, device):
    """Scratch for the K-slice partials of a per-point contraction (cl3d_workspace_bytes(CL3D_OP_POINT_GEMM = 14 /
    CL3D_OP_CONV1X1 = 15)): (tensor, bytes); the tensor is kept alive by the caller until its launches are queued."""
    nbytes = _lib.lib().cl3d_workspace_bytes(op, B, N, Co, 0, C)
    return torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=device), nbytes


def _build_inverse(idx, n_support):
    B = idx.shape[0]
    MK = idx[0].numel()
    off = torch.empty((B, n_support + 1), dtype=torch.int32, device=idx.device)
    slots = torch.empty((B, MK), dtype=torch.int32, device=idx.device)
    lib = _lib.lib()
    ws_bytes = lib.cl3d_workspace_bytes(11, B, n_support, MK, 1, 0)  # CL3D_OP_INVERSE_INDEX
    ws = torch.empty((max(ws_bytes, 1),), dtype=torch.uint8, device=idx.device)
    with _lib.on_device(idx.device):
        _lib.check(lib.cl3d_build_inverse_index(_p(idx), B, n_support, MK, _p(off), _p(slots), _p(ws), ws_bytes,
                                                _stream(idx)))
    return off, slots


def inverse_index(idx, n_support, prefetch=False, after=None):
    """CSR inverse of idx [B,M,K] (or [B,M]) -> (off [B,N+1], slots [B,MK]); memoised on the tensor.

    prefetch=True (forward pass, when a backward will follow): start the build on the index stream right
    behind the ball query and return nothing; the later call waits for it.  `after` (an event, prefetch only): the
    build also waits for it -- a dependency it does not need, used to ORDER the side queue of a captured step
    (_start_inverse)."""
    cached = getattr(idx, '_cl3d_inverse', None)
    if cached is not None and cached[0] == n_support:
        if not prefetch and cached[3] is not None:
            torch.cuda.current_stream(idx.device).wait_event(cached[3])
            idx._cl3d_inverse = cached[:3] + (None,)
            _PENDING[:] = [t for t in _PENDING if t is not idx]
        return cached[1], cached[2]
    ev = None
    if prefetch and pt_utils.async_index():
        main, side = torch.cuda.current_stream(idx.device), pt_utils.index_stream(idx.device, 1)
        if getattr(idx, '_cl3d_ready', None) is None:
            side.wait_stream(main)  # an idx produced in line on the caller's stream: the build follows it
        with torch.cuda.stream(side):
            wait_ready(idx)  # the ball query ran on the index stream; this covers a cached idx too
            if after is not None:
                side.wait_event(after)
            if not torch.cuda.is_current_stream_capturing():
                idx.record_stream(side)  # read here: the allocator must not recycle it before this stream is done
            off, slots = _build_inverse(idx, n_support)
            ev = torch.cuda.Event()
            ev.record(side)
        if not torch.cuda.is_current_stream_capturing():
            off.record_stream(main)
            slots.record_stream(main)
    elif prefetch:
        return None
    else:
        wait_ready(idx)
        off, slots = _build_inverse(idx, n_support)
    idx._cl3d_inverse = (n_support, off, slots, ev)
    return off, slots


def _join_inverse(idx):
    """End of a forward pass: the caller's stream picks up the CSR build that ran beside it (joined in the same
    thread that forked it; the backward then finds a finished table).  (Leaving the join to the backward pass's first
    consumer -- so that the first backward kernel does not inherit a cross-queue dependency -- was measured in round 3:
    0.3754 against 0.3702 ms per replayed step, no gain.)"""
    cached = getattr(idx, '_cl3d_inverse', None)
    if cached is not None and cached[3] is not None:
        torch.cuda.current_stream(idx.device).wait_event(cached[3])
        idx._cl3d_inverse = cached[:3] + (None,)
        _PENDING[:] = [t for t in _PENDING if t is not idx]


# Bookkeeping of ONE stream capture at a time per process (the engine runs one process per GPU).  Plain module state on
# purpose: a step's backward pass runs on the autograd engine's device thread, not on the thread that captures, and has to
# find -- and clear -- what the forward pass left here.
_WHOLE_STEP = [False]
_PENDING = []  # idx tensors whose CSR build a captured forward left on the index stream for its backward to join
_FORKS = [None]  # None: gradient products fork in a declared whole-step capture only; else forked_gradients()


def _forks_allowed():
    return _WHOLE_STEP[0] if _FORKS[0] is None else _FORKS[0]


@contextlib.contextmanager
def forked_gradients(on):
    """Explicit override of where the two gradient products of a contraction may run side by side on two HIP streams
    (_fork_join).  Default (no override): only inside a capture declared with whole_step_capture() -- a graph that holds
    forward AND backward.  forked_gradients(True) extends it to the capture it wraps: exact, as measured, for a graph
    that holds the forward pass and PART of its backward (scripts/bench_backbone.py --overlap, graph A);
    a graph that holds a backward pass ALONE gave wrong, replay-varying gradients with such forks in it whenever the
    warm-up had run on a stream of its own (autograd's AccumulateGrad stream then is a third branch of that graph)
    -- use it there only with warm-up and capture on closerlook3d_amd.step_stream().  forked_gradients(False) takes the
    forks out of a declared whole-step capture."""
    old = _FORKS[0]
    _FORKS[0] = None if on is None else bool(on)
    try:
        yield
    finally:
        _FORKS[0] = old


_STEP_STREAMS = {}


def step_stream(device=None):
    """One stream per device for warm-up AND capture of a training step (closerlook3d_amd.step_stream has the why)."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    st = _STEP_STREAMS.get(dev)
    if st is None:
        st = _STEP_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return st


def _is_unjoined_capture_error(e):
    text = f"{type(e).__name__}: {e}".lower()
    return "unjoined" in text or "capture" in text and "join" in text


@contextlib.contextmanager
def whole_step_capture(on=True):
    """Declare that the HIP graph being captured holds a whole training step -- every forward pass TOGETHER with its
    backward pass (bench.py, scripts/bench_backbone.py, a captured training loop).  The PointWiseMLP's forward pass may
    then leave the CSR build it forked for its backward (on the index stream) to be joined by that backward's
    support-major pass, inside the same capture: joined at the end of the forward, the FIRST backward kernel inherits
    a cross-queue wait for a table only a later kernel reads (replayed step, round 3: ~12 us between the forward's
    last kernel and the backward's first, against ~5 us between kernels of one queue); and the two gradient products of
    a contraction may run side by side (_fork_join).  Without the declaration a captured forward pass ends fully joined,
    as hipStreamEndCapture demands of a capture that stops there (torch.cuda.make_graphed_callables captures forward and
    backward separately), and a captured backward pass is single-stream; the kernels are the same either way.  Eager
    launches never need it.  A declared capture whose backward did NOT run inside it is reported here by name instead of
    as a bare hipErrorStreamCaptureUnjoined; any OTHER error raised inside the context (out of memory, a kernel's error
    code) reaches the caller unchanged."""
    old = _WHOLE_STEP[0]
    _WHOLE_STEP[0] = bool(on)
    del _PENDING[:]
    failed = False
    try:
        yield
    except Exception as e:
        failed = True
        if _PENDING and _is_unjoined_capture_error(e):
            raise RuntimeError(
                f"whole_step_capture(): {len(_PENDING)} PointWiseMLP forward pass(es) left their CSR build on the index "
                "stream, but no backward pass joined it inside the capture -- capture forward AND backward together, "
                "or drop the declaration") from e
        raise
    finally:
        _WHOLE_STEP[0] = old
        left = len(_PENDING)
        del _PENDING[:]
        if left and not failed:
            raise RuntimeError(f"whole_step_capture(): {left} PointWiseMLP forward pass(es) were captured without their "
                               "backward pass; the captured graph ends with unjoined work")


def _join_geometry(idx):
    """End of a PointWiseMLP forward pass: the caller's stream picks up the CSR build that ran beside it -- except in a
    declared whole-step capture, where the backward's support-major pass (the table's only reader in this operator)
    joins it right before it runs (whole_step_capture).  In eager mode the join is always taken: it costs nothing
    there, and the backward then finds a finished table whatever mode IT runs in."""
    cached = getattr(idx, '_cl3d_inverse', None)
    if cached is not None and cached[3] is not None and _WHOLE_STEP[0] and torch.cuda.is_current_stream_capturing():
        if not any(t is idx for t in _PENDING):
            _PENDING.append(idx)
        return
    _join_inverse(idx)


def _transposed(t):
    """[B,R,C] -> contiguous [B,C,R] through the engine's tiled transpose (channel-major <-> point-major rows)."""
    t = t.contiguous()
    B, R, C = t.shape
    out = torch.empty((B, C, R), dtype=t.dtype, device=t.device)
    with _lib.on_device(t.device):
        _lib.check(_lib.lib().cl3d_transpose(_p(t), B, R, C, _p(out), _stream(t)))
    return out


class _FusedReduce(Function):
    """out[b,c,j] = reduce_k w_c(rel) * mask * f[b,c,idx]  (PosPool / AdaptiveWeight / PseudoGrid)."""

    @staticmethod
    def forward(ctx, features, p0, p1, op, query_xyz, support_xyz, query_mask, idx, idx_mask, radius,
                normalize, reduction, pint, pfloat, constant, need_grad, defer_join=False):
        B, C, N = features.shape
        _, M, K = idx.shape
        ft = _transposed(features)
        pre = _mark(features.device) if need_grad else None
        if need_grad and REDUCE_CSR_FIRST and pt_utils._BQ_CACHE is None:
            _start_inverse(idx, N, None)  # the build captured before the gather pass (stand-alone operator: pointwise_mlp's note)
        wait_ready(idx)  # ball query ran on the index stream
        out = torch.empty((B, C, M), dtype=torch.float32, device=features.device)  # channel-major, written by the kernel
        slotrec = torch.empty((B, M, K, 4), dtype=torch.float32, device=features.device) if need_grad else None
        pairs = None
        if need_grad and op == OP_PSEUDOGRID and C % 4 == 0 and not constant:  # the slots' non-zero influences, kept for the backward
            pairs = torch.empty((B, M, K, 8), dtype=torch.float32, device=features.device)
        with _lib.on_device(features.device):
            _lib.check(_lib.lib().cl3d_fused_reduce_fwd(
                op, _p(query_xyz), _p(support_xyz), _p(query_mask), _p(idx), _p(idx_mask), _p(ft), B, N, M, K, C,
                float(radius), int(normalize), reduction, _p(p0), _p(p1), pint, float(pfloat), int(constant),
                _p(out), 1, _p(slotrec), _p(pairs), _stream(features)))
        if need_grad:
            _start_inverse(idx, N, pre)
        ctx.save_for_backward(ft, slotrec, p0, p1, pairs)
        ctx.idx = idx
        ctx.meta = (op, B, N, M, K, C, pint, pfloat, constant)
        if not defer_join:
            _join_inverse(idx)
        return out

    @staticmethod
    def backward(ctx, gout):
        ft, slotrec, p0, p1, pairs = ctx.saved_tensors
        op, B, N, M, K, C, pint, pfloat, constant = ctx.meta
        gout_t = _transposed(gout)
        off, slots = inverse_index(ctx.idx, N)
        dfeat = torch.empty((B, C, N), dtype=torch.float32, device=gout.device)  # channel-major, written by the kernel
        lib = _lib.lib()
        nparts = lib.cl3d_fused_param_partials(op, B, N, C)
        npar = {OP_ADAPTIVE: 4, OP_PSEUDOGRID: 16}.get(op, 0)
        dparam = torch.empty((nparts, C, npar), dtype=torch.float32, device=gout.device) if nparts else None
        with _lib.on_device(gout.device):
            _lib.check(lib.cl3d_fused_reduce_bwd(op, _p(gout_t), _p(ft), _p(slotrec), _p(pairs), _p(ctx.idx), _p(off), _p(slots), B, N, M, K,
                                                 C, _p(p0), _p(p1), pint, float(pfloat), int(constant), _p(dfeat), 1,
                                                 _p(dparam), nparts, _stream(gout)))
        g0 = g1 = None
        # the blocks' partial sums are added in double, in a fixed order: they are a few hundred terms that cancel heavily
        # (a channel's gradient is often two orders of magnitude below its partial sums); one launch that also writes the
        # parameters' own layouts (the same through five or six library launches: 33 us of the replayed PseudoGrid step)
        if op == OP_ADAPTIVE:
            g0 = torch.empty((C // pint, 3), dtype=torch.float32, device=gout.device)
            g1 = torch.empty((C // pint,), dtype=torch.float32, device=gout.device)
        elif op == OP_PSEUDOGRID:
            g1 = torch.empty((pint, C), dtype=torch.float32, device=gout.device)
        if g1 is not None:
            with _lib.on_device(gout.device):
                _lib.check(lib.cl3d_fused_param_reduce(op, _p(dparam), nparts, C, pint, _p(g0), _p(g1), _stream(gout)))
        return (dfeat, g0, g1) + (None,) * 14


def _checked(features, query_xyz, support_xyz, query_mask, support_mask):
    """The fused kernels read raw pointers: apply the reference's CHECK_IS_FLOAT / CHECK_IS_INT / CHECK_CONTIGUOUS rules
    (utils.h:10-30) here as `_ext` does -- float32 / int32 on the GPU, same device, made contiguous where a copy is
    legitimate (features, coordinates), refused otherwise -- so half / double features under autocast or a sliced xyz
    raise instead of reading out of bounds."""
    from ._ext import _check, _check_dev
    features, query_xyz, support_xyz = features.contiguous(), query_xyz.contiguous(), support_xyz.contiguous()
    query_mask, support_mask = query_mask.contiguous(), support_mask.contiguous()
    _check("support_features", features, torch.float32)
    _check("query_xyz", query_xyz, torch.float32)
    _check("support_xyz", support_xyz, torch.float32)
    _check("query_mask", query_mask, torch.int32)
    _check("support_mask", support_mask, torch.int32)
    _check_dev(features, query_xyz=query_xyz, support_xyz=support_xyz, query_mask=query_mask, support_mask=support_mask)
    return features, query_xyz, support_xyz, query_mask, support_mask


def _wants_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample, need_grad=False):
    """Ball query on the index stream; the fused Functions wait_ready() the result right before their first kernel that
    reads it and start the CSR inverse (when a backward will follow) right BEHIND that kernel (_start_inverse; `need_grad`
    is kept for the callers' signature: the build is no longer started here).  Inside a backbone's forward, where the
    query ran ahead of the feature pass, starting the build right behind the QUERY instead was measured again in round 5
    (same box, alternating runs): config 2 8.09 against 8.12 ms (noise), config 3 6.19 against 5.68 ms, config 4 6.49
    against 6.03 ms -- the build behind the consumer wins or ties everywhere."""
    idx, idx_mask = _ball_query(query_xyz.contiguous(), support_xyz.contiguous(), query_mask.contiguous(),
                                support_mask.contiguous(), radius, nsample, defer=True)
    return idx, idx_mask


def _mark(device):
    """An event recorded on the caller's stream at this point -- inside a capture an ordering edge for the index stream's
    work (_start_inverse), in eager launches a plain (cheap) event with the same meaning; None without the index streams."""
    if not (device.type == 'cuda' and pt_utils.async_index()):
        return None
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    return ev


def _start_inverse(idx, n_support, after=None):
    """Fork the CSR build of idx onto the index stream (no-op when it exists or is under way).  Called right AFTER the
    forward kernel that reads idx has been enqueued, with `after` = _mark() taken right BEFORE that kernel.

    Why the order matters (scripts/micro/graph_queues.hip replays graphs of the step's shape built from timed spin
    kernels in every capture order; gpurun_out/r05b, DESIGN 3.2 "Round 5"): the HIP runtime lays a graph out on TWO
    queues, depth-first from the roots in capture order; a node's FIRST-captured dependent stays on the node's queue,
    the next one goes to the other queue, and so does the next root.  A queue runs its nodes in that visiting order.
    The step has two roots -- the ball query and the feature-side preparation (weight split + per-point product, or the
    layout change) -- and the query has two dependents: the gather pass (critical: everything follows it) and the CSR
    build (~100 us of slack).  So: (1) the gather pass must be captured BEFORE the CSR build, or it -- and the whole
    chain behind it -- pays a cross-queue hand-over behind the query (r4: 12.7 us, again 10 us in front of the
    support-major pass); (2) the CSR build then shares the side queue with the feature-side preparation and is VISITED
    FIRST: the preparation, which the gather pass needs, would sit behind a build that waits for the query
    (measured: step 0.314 -> 0.374 ms).  Making the build wait for `after` -- the preparation, finished long before the
    query -- forces the side queue's order: preparation, then build.

    That is the order inside a backbone (shared, mostly prefetched geometry).  A STAND-ALONE operator step takes the other
    one since the round's second session -- build captured BEFORE the gather pass, `after` = None (pointwise_mlp,
    REDUCE_CSR_FIRST): point (1) above prices the gather pass's hand-over behind the query, but the gather pass pays one
    hand-over anyway (behind the feature-side preparation), and with the build as the query's first dependent its count pass
    runs alone the moment the query ends instead of squeezed in behind the gather pass's workgroups (60 -> 11 us; step
    0.2885 -> 0.281 ms, profiles/r05/pw_capture_order.txt)."""
    inverse_index(idx, n_support, prefetch=True, after=after)


def _deferred(out, idx, defer_join):
    """defer_join: the operator module runs its BatchNorm + ReLU behind this call and joins the CSR build after those
    (join_pending) instead of before them -- the build is longer than the gather pass it runs beside (replayed PosPool
    step: the statistics pass started 19 us after the gather pass ended, waiting for a table only the backward reads)."""
    if defer_join:
        out._cl3d_pending = idx
    return out


def join_pending(out):
    """The caller's stream picks up the CSR build left running by an operator called with defer_join=True."""
    idx = getattr(out, '_cl3d_pending', None)
    if idx is not None:
        _join_inverse(idx)
        out._cl3d_pending = None


def pospool(query_xyz, support_xyz, query_mask, support_mask, features, radius, nsample, embedding, reduction,
            defer_join=False, out_bn=None):
    features, query_xyz, support_xyz, query_mask, support_mask = _checked(features, query_xyz, support_xyz, query_mask, support_mask)
    C = features.shape[1]
    if embedding == 'xyz':
        if C % 3:
            raise RuntimeError(f"PosPool xyz needs C % 3 == 0, got {C}")
        op, p0 = OP_POSPOOL_XYZ, None
    else:
        if C % 6:
            raise RuntimeError(f"PosPool sin_cos needs C % 6 == 0, got {C}")
        fd = C // 6
        op = OP_POSPOOL_SINCOS
        p0 = torch.pow(1.0 * 1000, (1.0 / fd) * torch.arange(fd, dtype=torch.float32, device=features.device))
    if _use_reduce_pass(_wants_grad(features)):
        return _reduce_pass(features, p0, None, op, query_xyz, support_xyz, query_mask, support_mask, radius, nsample, True,
                            _RED[reduction], 0, 0.0, False, out_bn)
    idx, idx_mask = _query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample, _wants_grad(features))
    out = _FusedReduce.apply(features, p0, None, op, query_xyz, support_xyz, query_mask, idx, idx_mask, radius,
                             True, _RED[reduction], 0, 0.0, False, _wants_grad(features), defer_join)
    return _deferred(out, idx, defer_join)


def _use_reduce_pass(need_grad):
    """The one-call-per-pass path of the three gather-and-reduce operators (pass_calls._ReducePass): a stand-alone operator
    whose backward will follow, launched eagerly, no per-forward geometry memo -- the conditions of _use_pass_calls."""
    return (PASS_CALLS and need_grad and pt_utils._BQ_CACHE is None and pt_utils.ASYNC_INDEX == 'auto'
            and not torch.cuda.is_current_stream_capturing())


def _reduce_pass(features, p0, p1, op, query_xyz, support_xyz, query_mask, support_mask, radius, nsample, normalize,
                 reduction, pint, pfloat, constant, out_bn=None):
    """out_bn (an nn.BatchNorm1d in training mode that the kernels cover, or None): the operator's BatchNorm + ReLU output
    transform inside the same two C-ABI calls; the result then is marked so that the module does not apply it again."""
    from .pass_calls import _ReducePass
    if out_bn is not None and _bn_unit_ok(features, out_bn) and out_bn.training and out_bn.running_mean is not None:
        out = _ReducePass.apply(features, p0, p1, op, query_xyz, support_xyz, query_mask, support_mask, radius, nsample,
                                normalize, reduction, pint, pfloat, constant, out_bn.weight, out_bn.bias,
                                out_bn.running_mean, out_bn.running_var, _step_counter(out_bn), out_bn.momentum, out_bn.eps)
        out._cl3d_activated = True
        return out
    return _ReducePass.apply(features, p0, p1, op, query_xyz, support_xyz, query_mask, support_mask, radius, nsample,
                             normalize, reduction, pint, pfloat, constant)


def adaptive_weight(query_xyz, support_xyz, query_mask, support_mask, features, radius, nsample, mlps,
                    shared_channels, reduction, defer_join=False, out_bn=None):
    features, query_xyz, support_xyz, query_mask, support_mask = _checked(features, query_xyz, support_xyz, query_mask, support_mask)
    conv = mlps.conv0
    w = conv.weight.view(conv.weight.shape[0], 3)
    if _use_reduce_pass(_wants_grad(features, w, conv.bias)):
        return _reduce_pass(features, w, conv.bias, OP_ADAPTIVE, query_xyz, support_xyz, query_mask, support_mask, radius,
                            nsample, True, _RED[reduction], int(shared_channels), 0.0, False, out_bn)
    idx, idx_mask = _query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample,
                           _wants_grad(features, w, conv.bias))
    out = _FusedReduce.apply(features, w, conv.bias, OP_ADAPTIVE, query_xyz, support_xyz, query_mask, idx,
                             idx_mask, radius, True, _RED[reduction], int(shared_channels), 0.0, False,
                             _wants_grad(features, w, conv.bias), defer_join)
    return _deferred(out, idx, defer_join)


def pseudo_grid(query_xyz, support_xyz, query_mask, support_mask, features, radius, nsample, k_points,
                kernel_weights, extent, influence, defer_join=False, out_bn=None):
    features, query_xyz, support_xyz, query_mask, support_mask = _checked(features, query_xyz, support_xyz, query_mask, support_mask)
    if _use_reduce_pass(_wants_grad(features, kernel_weights)):
        return _reduce_pass(features, k_points.contiguous(), kernel_weights, OP_PSEUDOGRID, query_xyz, support_xyz, query_mask,
                            support_mask, radius, nsample, False, _RED['sum'], int(k_points.shape[0]), 1.0 / float(extent),
                            influence == 'constant', out_bn)
    idx, idx_mask = _query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample,
                           _wants_grad(features, kernel_weights))
    out = _FusedReduce.apply(features, k_points.contiguous(), kernel_weights, OP_PSEUDOGRID, query_xyz,
                             support_xyz, query_mask, idx, idx_mask, radius, False, _RED['sum'],
                             int(k_points.shape[0]), 1.0 / float(extent), influence == 'constant',
                             _wants_grad(features, kernel_weights), defer_join)
    return _deferred(out, idx, defer_join)


class _MaxPool(Function):
    """out[b,c,j] = max_k f[b,c,idx[b,j,k]] with max_pool2d's gradient routing (first maximum).

    Round 6: the forward pass keeps the arg-max's SUPPORT INDEX per (channel, query) (`cl3d_maxpool_fwd_targets`) and the
    backward is a scatter with one target per (query, channel) on the channel-major gradient (`cl3d_maxpool_bwd_targets`:
    doubles in LDS, exact and order-free) -- no CSR inverse of idx, no transposed copy of the gradient.  MAXPOOL_TARGETS =
    False restores the slot-byte form with its ordered gather through the CSR inverse (the A/B arm; same values up to the
    rounding of a sum of a few floats)."""

    @staticmethod
    def forward(ctx, features, idx, need_grad):
        B, C, N = features.shape
        _, M, K = idx.shape
        ft = _transposed(features)
        targets = need_grad and MAXPOOL_TARGETS
        pre = _mark(features.device) if need_grad and not targets else None
        wait_ready(idx)
        out = torch.empty((B, C, M), dtype=torch.float32, device=features.device)
        kept = None
        with _lib.on_device(features.device):
            if targets:
                kept = torch.empty((B, C, M), dtype=torch.int32, device=features.device)
                _lib.check(_lib.lib().cl3d_maxpool_fwd_targets(_p(idx), _p(ft), B, N, M, K, C, _p(out), _p(kept),
                                                               _stream(features)))
            else:
                kept = torch.empty((B, M, C), dtype=torch.uint8, device=features.device) if need_grad else None
                _lib.check(_lib.lib().cl3d_maxpool_fwd(_p(idx), _p(ft), B, N, M, K, C, _p(out), _p(kept),
                                                       _stream(features)))
        if need_grad and not targets:
            _start_inverse(idx, N, pre)
        ctx.save_for_backward(kept)
        ctx.idx = None if targets else idx
        ctx.targets = targets
        ctx.meta = (B, N, M, K, C)
        if not targets:
            _join_inverse(idx)
        return out

    @staticmethod
    def backward(ctx, gout):
        (kept,) = ctx.saved_tensors
        B, N, M, K, C = ctx.meta
        dfeat = torch.empty((B, C, N), dtype=torch.float32, device=gout.device)  # channel-major, written by the kernel
        if ctx.targets:
            gout = gout.contiguous()
            with _lib.on_device(gout.device):
                _lib.check(_lib.lib().cl3d_maxpool_bwd_targets(_p(gout), _p(kept), B, N, M, C, _p(dfeat), _stream(gout)))
            return dfeat, None, None
        gout_t = _transposed(gout)
        off, slots = inverse_index(ctx.idx, N)
        with _lib.on_device(gout.device):
            _lib.check(_lib.lib().cl3d_maxpool_bwd(_p(gout_t), _p(kept), _p(off), _p(slots), B, N, M, K, C, _p(dfeat), 1,
                                                   _stream(gout)))
        return dfeat, None, None


def max_pool(query_xyz, support_xyz, query_mask, support_mask, features, radius, nsample):
    """MaskedMaxPool's pooling step on the fused path (nsample <= 255)."""
    features, query_xyz, support_xyz, query_mask, support_mask = _checked(features, query_xyz, support_xyz, query_mask, support_mask)
    idx, _ = _query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample, _wants_grad(features))
    return _MaxPool.apply(features.contiguous(), idx, _wants_grad(features))


class _BnRelu(Function):
    """ReLU(BatchNorm1d(x)) on channel-major x [B,C,N]; csrc/bn_relu.hip.  Training: one launch each way when a channel
    has few values (deep stages), statistics + apply passes otherwise; x and the output are kept for backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, num_batches_tracked=None):
        x = x.contiguous()
        B, C, N = x.shape
        dev = x.device
        lib = _lib.lib()
        out = torch.empty_like(x)
        with _lib.on_device(dev):
            st = _stream(x)
            if training:
                vec = torch.empty((4, C), dtype=torch.float32, device=dev)
                nparts = lib.cl3d_bn_partials(B, C, N)
                partial = torch.empty((nparts, C, 2), dtype=torch.float64, device=dev)
                _lib.check(lib.cl3d_bn_add_relu_train_fwd(_p(x), _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                                          _p(num_batches_tracked), float(eps), float(momentum), None, None,
                                                          None, None, None, None, 0.0, 0.0, 1, B, C, N, _p(partial), nparts,
                                                          _p(vec), None, _p(out), st))
                ctx.save_for_backward(x, out, vec, gamma)
                ctx.meta = (B, C, N, nparts)
            else:
                invstd64 = torch.rsqrt(running_var.double() + eps)
                scale64 = gamma.double() * invstd64
                scale = scale64.float()
                shift = (beta.double() - running_mean.double() * scale64).float()
                _lib.check(lib.cl3d_bn_relu_apply(_p(x), _p(scale), _p(shift), B, C, N, _p(out), st))
        ctx.training = training
        return out

    @staticmethod
    def backward(ctx, g):
        if not ctx.training:
            raise NotImplementedError("fused BatchNorm+ReLU backward needs training-mode statistics")
        x, out, vec, gamma = ctx.saved_tensors
        B, C, N, nparts = ctx.meta
        g = g.contiguous()
        dev = g.device
        dx = torch.empty_like(x)
        coef = torch.empty((5, C), dtype=torch.float32, device=dev)
        partial = torch.empty((2, nparts, C, 2), dtype=torch.float64, device=dev)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().cl3d_bn_add_relu_bwd(_p(g), _p(out), _p(x), _p(vec[2]), _p(vec[3]), _p(gamma), None, None,
                                                      None, None, 1, B, C, N, float(B * N), _p(partial), nparts, _p(coef),
                                                      None, _p(dx), None, _stream(g)))
        return dx, coef[3], coef[4], None, None, None, None, None, None


def _step_counter(bn):
    """nn.BatchNorm's num_batches_tracked when the engine can bump it inside the statistics kernel (an int64 scalar on
    the module's device, as PyTorch registers it); anything else is bumped here and None is returned."""
    t = getattr(bn, 'num_batches_tracked', None)
    if t is None:
        return None
    if t.dtype == torch.int64 and t.is_cuda and t.numel() == 1:
        return t
    t.add_(1)
    return None


def bn_relu(x, bn):
    """The engine's BatchNorm1d + ReLU with an nn.BatchNorm1d module's parameters, buffers and semantics; returns
    None when the configuration is outside what the kernels cover (the caller then runs the nn modules)."""
    training = bn.training or bn.running_mean is None
    if (not x.is_cuda or x.dtype != torch.float32 or x.dim() != 3 or not bn.affine or bn.momentum is None
            or not bn.track_running_stats or (not training and torch.is_grad_enabled() and
                                              _wants_grad(x, bn.weight, bn.bias))):
        return None
    # num_batches_tracked is bumped by the kernel that updates the running statistics (no launch of its own)
    return _BnRelu.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.momentum, bn.eps,
                         _step_counter(bn) if training else None)


class _PointwiseMLP(Function):
    """max_k ReLU(BN(W_r rel + H[centre] + G[nbr])) on point-major rows; see csrc/fused_pwmlp.hip."""

    @staticmethod
    def forward(ctx, ght, wr, gamma, beta, running_mean, running_var, query_xyz, support_xyz, idx, radius,
                training, momentum, eps, need_grad, num_batches_tracked=None, rows_out=False):
        """rows_out (training only): return (ystar [B,M,Co], scale [Co], shift [Co]) instead of the activated
        channel-major tensor -- the consumer (conv2 of a bottleneck, _Conv1x1Rows) applies max(scale * y + shift, 0)
        while it stages its operand, and hands back the gradient with respect to the activated rows."""
        B, N, two_co = ght.shape
        Co = two_co // 2
        _, M, K = idx.shape
        dev = ght.device
        lib = _lib.lib()
        n = B * M * K
        nparts = lib.cl3d_pwmlp_partials(B, M, Co)
        pre = _mark(dev) if (need_grad and training) else None
        wait_ready(idx)  # ball query ran on the index stream while the per-point GEMM ran here
        out = torch.empty((B, Co, M), dtype=torch.float32, device=dev)  # channel-major, written by the kernels
        with _lib.on_device(dev):
            st = _stream(ght)
            if training:
                # one gather pass: batch statistics AND, per (query, channel), the pre-activation that wins the
                # max, its slot and sum_k y -- the rest of forward and most of backward is algebra on those
                vec = torch.empty((4, Co), dtype=torch.float32, device=dev)
                scale, shift, mean, invstd = vec[0], vec[1], vec[2], vec[3]
                ystar = torch.empty((B, M, Co), dtype=torch.float32, device=dev)  # (an output when rows_out)
                sy = torch.empty((B, M, Co), dtype=torch.float32, device=dev)
                kstar = torch.empty((B, M, Co), dtype=torch.uint8, device=dev)
                partial = torch.empty((nparts, Co, 8), dtype=torch.float64, device=dev)
                sums = torch.empty((Co, 6), dtype=torch.float64, device=dev)
                _lib.check(lib.cl3d_pwmlp_stats(_p(query_xyz), _p(support_xyz), _p(idx), _p(ght), _p(wr), _p(gamma),
                                                B, N, M, K, Co, float(radius), _p(ystar), _p(kstar), _p(sy),
                                                _p(partial), nparts, st))
                if need_grad:
                    # (the build made to wait for the END of the statistics pass, or captured behind the activation pass,
                    # was measured too: 0.302-0.304 / 0.311 ms against 0.301 ms -- the contention moves, it does not go)
                    _start_inverse(idx, N, pre)
                # batch statistics, scale/shift and the running-statistics update in one small launch
                _lib.check(lib.cl3d_pwmlp_finalize_stats(_p(partial), nparts, Co, float(n), float(eps), float(momentum),
                                                         _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                                         _p(num_batches_tracked), _p(scale), _p(shift), _p(mean),
                                                         _p(invstd), _p(sums), st))
                if not rows_out:
                    _lib.check(lib.cl3d_pwmlp_apply(_p(ystar), _p(scale), _p(shift), B, M, Co, _p(out), st))
                if need_grad:
                    ctx.save_for_backward(ght, wr, gamma, vec, ystar, sy, kstar, sums, query_xyz, support_xyz)
                    ctx.radius = float(radius)
                    ctx.idx = idx
                    ctx.meta = (B, N, M, K, Co, nparts)
                ctx.rows_out = bool(rows_out)
                if rows_out:
                    _join_geometry(idx)
                    ctx.mark_non_differentiable(scale, shift)
                    return ystar, scale, shift  # ystar is saved above AND returned: an output may be saved
            else:
                if need_grad:
                    raise NotImplementedError("fused PointWiseMLP backward needs training-mode BatchNorm")
                invstd64 = torch.rsqrt(running_var.double() + eps)
                scale64 = gamma.double() * invstd64
                scale = scale64.float()
                shift = (beta.double() - running_mean.double() * scale64).float()
                _lib.check(lib.cl3d_pwmlp_fwd(_p(query_xyz), _p(support_xyz), _p(idx), _p(ght), _p(wr), _p(scale),
                                              _p(shift), B, N, M, K, Co, float(radius), _p(out), 1, None, None, st))
        _join_geometry(idx)
        return out

    @staticmethod
    def backward(ctx, gout, *unused):
        ght, wr, gamma, vec, ystar, sy, kstar, sums, query_xyz, support_xyz = ctx.saved_tensors
        B, N, M, K, Co, nparts = ctx.meta
        idx = ctx.idx
        dev = gout.device
        lib = _lib.lib()
        n = B * M * K
        gout = gout.contiguous()  # channel-major [B,Co,M] (or point-major rows [B,M,Co]), read directly by the kernel
        gout_cm = 0 if getattr(ctx, 'rows_out', False) else 1
        with _lib.on_device(dev):
            st = _stream(gout)
            dz_cm = torch.empty((B, Co, M), dtype=torch.float32, device=dev)
            ts_cm = torch.empty((B, Co, M), dtype=torch.int32, device=dev)
            partial = torch.empty((nparts, Co, 8), dtype=torch.float64, device=dev)
            # for the support-major pass: dz again as point-major rows, and one 16-byte record per query
            # {coordinates, centre idx[j, 0]} (an entry of that pass then costs one L2 request instead of three)
            dz_t = torch.empty((B, M, Co), dtype=torch.float32, device=dev)
            qtab = torch.empty((B, M, 4), dtype=torch.float32, device=dev)
            _lib.check(lib.cl3d_pwmlp_bwd_rows(_p(gout), gout_cm, _p(ystar), _p(kstar), _p(idx), _p(query_xyz), _p(support_xyz),
                                               ctx.radius, _p(vec[0]), _p(vec[1]), _p(vec[2]), _p(vec[3]), B, N, M, K, Co,
                                               _p(dz_cm), _p(ts_cm), _p(dz_t), _p(qtab), _p(partial), nparts, st))
            hit = torch.empty((B, Co, N), dtype=torch.float32, device=dev)
            # d gamma, d beta, d W_r and the coefficients of  dy = A dz + Bc + D y  (BatchNorm backward is affine in y)
            coef = torch.empty((5, Co), dtype=torch.float32, device=dev)
            cA, cB, cD, dgamma, dbeta = coef[0], coef[1], coef[2], coef[3], coef[4]
            dwr = torch.empty((Co, 3), dtype=torch.float32, device=dev)

            # the arg-max scatter and the per-channel algebra -- both only need what bwd_rows left -- in one launch (as
            # two launches they are 10.6 + 6.5 us one after the other; a fork/join across HIP streams costs more idle time
            # in a captured graph than either takes)
            _lib.check(lib.cl3d_pwmlp_bwd_hits_coeffs(_p(partial), nparts, float(n), _p(gamma), _p(vec[2]), _p(vec[3]), _p(sums),
                                                      _p(cA), _p(cB), _p(cD), _p(dgamma), _p(dbeta), _p(dwr), _p(dz_cm),
                                                      _p(ts_cm), B, N, M, Co, _p(hit), st))
            dght = torch.empty((B, N, 2 * Co), dtype=torch.float32, device=dev)
            off, slots = inverse_index(idx, N)  # (waits for the build the forward pass forked, if it is still pending)
            _lib.check(lib.cl3d_pwmlp_bwd_support(_p(ght), _p(wr), _p(cA), _p(cB), _p(cD), _p(hit), _p(dz_t), _p(sy),
                                                  _p(qtab), _p(support_xyz), ctx.radius, _p(off), _p(slots),
                                                  B, N, M, K, Co, _p(dght), st))
        return (dght, dwr, dgamma, dbeta) + (None,) * 12


import os

PRECISIONS = {'f32': 0, 'bf16': 1}  # CL3D_PRECISION_*: arithmetic of the dense contraction only


class _PointRows(Function):
    """(features [B,C,N], W [Co,3+2C] = [W_r | W_c | W_d])  ->  ght [B,N,2Co] with rows [W_d f_i | (W_c - W_d) f_i]
    (once per point instead of once per (point, neighbour)), and W_r [Co,3].

    This is what is left of the reference's dense contraction (local_aggregation_operators.py:253-257,288-295) and
    it runs on the matrix cores through the engine's own kernel (csrc/mfma_gemm.hip: channel-major features in,
    point-major rows out, f32 or bf16 inputs with f32 accumulation); the backward pass is the same kernel twice
    (d features channel-major as the caller needs it; d W summed over all points in a fixed slice order and
    written in the Conv2d weight's own layout).
    """

    @staticmethod
    def forward(ctx, features, W, precision):
        features = features.contiguous()
        W = W.contiguous()
        B, C, N = features.shape
        Co = W.shape[0]
        dev = features.device
        wr = torch.empty((Co, 3), dtype=torch.float32, device=dev)
        wcat = torch.empty((2 * Co, C), dtype=torch.float32, device=dev)
        ght = torch.empty((B, N, 2 * Co), dtype=torch.float32, device=dev)
        ws, ws_bytes = _gemm_scratch(14, B, N, Co, C, dev)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().cl3d_pwmlp_point_gemm_fwd(_p(features), _p(W), B, C, N, Co, precision, _p(ght), _p(wr),
                                                            _p(wcat), _p(ws), ws_bytes, _stream(features)))
        ctx.save_for_backward(features, wcat)
        ctx.precision = precision
        ctx.w_leaf = _weight_leaf(W)
        return ght, wr

    @staticmethod
    def backward(ctx, dght, dwr):
        features, wcat = ctx.saved_tensors
        B, C, N = features.shape
        Co = wcat.shape[0] // 2
        dev = features.device
        lib = _lib.lib()
        dfeat = dW = None
        if dght is None:
            dght = torch.zeros((B, N, 2 * Co), dtype=torch.float32, device=dev)
        dght = dght.contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if need_x:
            dfeat = torch.empty((B, C, N), dtype=torch.float32, device=dev)
        if need_w:
            dW = torch.empty((Co, 3 + 2 * C), dtype=torch.float32, device=dev)
            dwr = dwr.contiguous() if dwr is not None else None
        prec = ctx.precision

        def data_grad():
            if need_x:
                ws, ws_bytes = _gemm_scratch(14, B, N, Co, C, dev)
                _lib.check(lib.cl3d_pwmlp_point_gemm_bwd_data(_p(dght), _p(wcat), B, C, N, Co, prec, _p(dfeat), _p(ws),
                                                              ws_bytes, _stream(features)))

        def weight_grad():
            if need_w:
                ws, ws_bytes = _gemm_scratch(14, B, N, Co, C, dev)
                _lib.check(lib.cl3d_pwmlp_point_gemm_bwd_weight(_p(features), _p(dght), _p(dwr), B, C, N, Co, prec, _p(dW),
                                                                _p(ws), ws_bytes, _stream(features)))

        with _lib.on_device(dev):
            if need_x and need_w and lib.cl3d_pwmlp_point_gemm_bwd_fused(B, C, N, Co, prec):
                # one kernel over d ght forms both (csrc/mfma_gemm.hip pwmlp_point_grads_kernel): nothing to fork
                ws, ws_bytes = _gemm_scratch(14, B, N, Co, C, dev)
                _lib.check(lib.cl3d_pwmlp_point_gemm_bwd(_p(features), None, None, _p(dght), _p(wcat), _p(dwr), B, C, N, Co,
                                                         prec, _p(dfeat), _p(dW), _p(ws), ws_bytes, _stream(features)))
            elif _fork_join(dev, weight_grad, data_grad, B * N,
                            (ctx.w_leaf, dW, (features, dght, dwr))):
                dW = None  # handed to the parameter by join_weight_gradients()
        return dfeat, dW, None


class _BnReluPointRows(Function):
    """The PointWiseMLP's per-point rows from a bottleneck's RAW conv1 output y1 [B,C,N]: BatchNorm (batch statistics)
    + ReLU are applied while the contraction stages its operand (cl3d_pwmlp_point_gemm_fwd_pro), so the activated
    tensor conv1 -> operator (backbones/resnet.py:32-34,54) is never written; backward: d act from the data-gradient
    product, the weight gradient again on the staged activation, then BatchNorm + ReLU backward on (d act, y1)."""

    @staticmethod
    def forward(ctx, y1, gamma, beta, bn, W, precision):
        y1 = y1.contiguous()
        W = W.contiguous()
        B, C, N = y1.shape
        Co = W.shape[0]
        dev = y1.device
        lib = _lib.lib()
        vec = torch.empty((4, C), dtype=torch.float32, device=dev)  # scale, shift, mean, invstd
        nparts = lib.cl3d_bn_partials(B, C, N)
        partial = torch.empty((nparts, C, 2), dtype=torch.float64, device=dev)
        wr = torch.empty((Co, 3), dtype=torch.float32, device=dev)
        wcat = torch.empty((2 * Co, C), dtype=torch.float32, device=dev)
        ght = torch.empty((B, N, 2 * Co), dtype=torch.float32, device=dev)
        ws, ws_bytes = _gemm_scratch(14, B, N, Co, C, dev)
        with _lib.on_device(dev):
            st = _stream(y1)
            _lib.check(lib.cl3d_bn_relu_stats(_p(y1), B, C, N, _p(partial), nparts, float(B * N), float(bn.eps),
                                              float(bn.momentum), _p(gamma), _p(beta), _p(bn.running_mean),
                                              _p(bn.running_var), _p(_step_counter(bn)), _p(vec[0]), _p(vec[1]),
                                              _p(vec[2]), _p(vec[3]), st))
            _lib.check(lib.cl3d_pwmlp_point_gemm_fwd_pro(_p(y1), _p(vec[0]), _p(vec[1]), _p(W), B, C, N, Co, precision,
                                                         _p(ght), _p(wr), _p(wcat), _p(ws), ws_bytes, st))
        ctx.save_for_backward(y1, vec, gamma, wcat)
        ctx.precision = precision
        ctx.nparts = nparts
        ctx.w_leaf = _weight_leaf(W)
        return ght, wr

    @staticmethod
    def backward(ctx, dght, dwr):
        y1, vec, gamma, wcat = ctx.saved_tensors
        B, C, N = y1.shape
        Co = wcat.shape[0] // 2
        dev = y1.device
        lib = _lib.lib()
        prec = ctx.precision
        if dght is None:
            dght = torch.zeros((B, N, 2 * Co), dtype=torch.float32, device=dev)
        dght = dght.contiguous()
        dwr = dwr.contiguous() if dwr is not None else None
        dact = torch.empty((B, C, N), dtype=torch.float32, device=dev)
        dW = torch.empty((Co, 3 + 2 * C), dtype=torch.float32, device=dev)

        def data_grad():
            ws, ws_bytes = _gemm_scratch(14, B, N, Co, C, dev)
            _lib.check(lib.cl3d_pwmlp_point_gemm_bwd_data(_p(dght), _p(wcat), B, C, N, Co, prec, _p(dact), _p(ws),
                                                          ws_bytes, _stream(y1)))

        def weight_grad():
            ws, ws_bytes = _gemm_scratch(14, B, N, Co, C, dev)
            _lib.check(lib.cl3d_pwmlp_point_gemm_bwd_weight_pro(_p(y1), _p(vec[0]), _p(vec[1]), _p(dght), _p(dwr), B, C, N,
                                                                Co, prec, _p(dW), _p(ws), ws_bytes, _stream(y1)))

        dy1 = torch.empty_like(y1)
        coef = torch.empty((5, C), dtype=torch.float32, device=dev)
        partial = torch.empty((ctx.nparts, C, 2), dtype=torch.float64, device=dev)
        with _lib.on_device(dev):
            if lib.cl3d_pwmlp_point_gemm_bwd_fused(B, C, N, Co, prec):
                ws, ws_bytes = _gemm_scratch(14, B, N, Co, C, dev)
                _lib.check(lib.cl3d_pwmlp_point_gemm_bwd(_p(y1), _p(vec[0]), _p(vec[1]), _p(dght), _p(wcat), _p(dwr), B, C, N,
                                                         Co, prec, _p(dact), _p(dW), _p(ws), ws_bytes, _stream(y1)))
            elif _fork_join(dev, weight_grad, data_grad, B * N, (ctx.w_leaf, dW, (y1, vec, dght, dwr))):
                dW = None  # handed to the parameter by join_weight_gradients()
            _lib.check(lib.cl3d_bn_relu_bwd(_p(dact), _p(y1), _p(vec[0]), _p(vec[1]), _p(vec[2]), _p(vec[3]), _p(gamma), B, C, N,
                                            float(B * N), _p(partial), ctx.nparts, _p(coef), _p(dy1), _stream(y1)))
        return dy1, coef[3], coef[4], None, dW, None


class _Conv1x1Rows(Function):
    """y [B,Co,N] = W max(scale * rows + shift, 0) with rows [B,N,C] the operator's point-major pre-activations and
    (scale, shift) its folded batch statistics: conv2 of a bottleneck reading the operator's output directly
    (backbones/resnet.py:56-58).  backward returns the gradient with respect to the ACTIVATED rows (the operator's own
    backward gates it by the ReLU and takes it through its BatchNorm) and d W."""

    @staticmethod
    def forward(ctx, rows, scale, shift, W, precision):
        rows = rows.contiguous()
        W = W.contiguous()
        B, N, C = rows.shape
        Co = W.shape[0]
        y = torch.empty((B, Co, N), dtype=torch.float32, device=rows.device)
        ws, ws_bytes = _gemm_scratch(15, B, N, Co, C, rows.device)
        with _lib.on_device(rows.device):
            _lib.check(_lib.lib().cl3d_conv1x1_rows_fwd(_p(rows), _p(scale), _p(shift), _p(W), B, C, N, Co, precision, _p(y),
                                                        _p(ws), ws_bytes, _stream(rows)))
        ctx.save_for_backward(rows, scale, shift, W)
        ctx.precision = precision
        ctx.w_leaf = _weight_leaf(W)
        return y

    @staticmethod
    def backward(ctx, dy):
        rows, scale, shift, W = ctx.saved_tensors
        B, N, C = rows.shape
        Co = W.shape[0]
        dev = rows.device
        dy = dy.contiguous()
        lib = _lib.lib()
        prec = ctx.precision
        drows = torch.empty_like(rows) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(W) if ctx.needs_input_grad[3] else None

        def data_grad():
            if drows is not None:
                ws, ws_bytes = _gemm_scratch(15, B, N, Co, C, dev)
                _lib.check(lib.cl3d_conv1x1_rows_bwd_data(_p(dy), _p(W), B, C, N, Co, prec, _p(drows), _p(ws), ws_bytes,
                                                          _stream(rows)))

        def weight_grad():
            if dW is not None:
                ws, ws_bytes = _gemm_scratch(15, B, N, Co, C, dev)
                _lib.check(lib.cl3d_conv1x1_rows_bwd_weight(_p(rows), _p(scale), _p(shift), _p(dy), B, C, N, Co, prec, _p(dW),
                                                            _p(ws), ws_bytes, _stream(rows)))

        with _lib.on_device(dev):
            if _fork_join(dev, weight_grad, data_grad, B * N,
                          (ctx.w_leaf, dW, (rows, scale, shift, dy))):
                dW = None  # handed to the parameter by join_weight_gradients()
        return drows, None, None, dW, None


def pointwise_bottleneck(conv1, la, conv2, shortcut, query_xyz, support_xyz, query_mask, support_mask, features,
                         identity, precision='f32'):
    """A whole PointWiseMLP bottleneck in training mode without the two [B,C,N] tensors between its layers
    (backbones/resnet.py:47-66):
        y1 = conv1(x)                          MFMA convolution, raw output
        ght = rows(act1(y1))                   BatchNorm + ReLU of conv1 inside the per-point contraction's staging
        rows, scale, shift = operator(ght)     point-major pre-activations, the operator's BatchNorm left to the consumer
        y2 = conv2(act(rows))                  conv2 reads the rows, the operator's BatchNorm + ReLU inside ITS staging
        out = ReLU(BN2(y2) + shortcut)         the fused tail of conv_bn_act
    conv1 / conv2 / shortcut: the bottleneck's `_conv_bn` units; la: its PointWiseMLP module.  Returns None when the
    configuration is outside what the kernels cover (the caller then runs layer by layer)."""
    c1, bn1 = conv1[0], conv1[1]
    c2, bn2 = conv2[0], conv2[1]
    mconv, mbn = la.mlps.conv0[0], la.mlps.conv0[1]
    if (la.reduction != 'max' or la.num_mlps != 1
            or la.feature_type != 'dp_fi_df' or not (bn1.training and bn2.training and mbn.training)):
        return None
    if c1.bias is not None or c1.kernel_size != (1,) or c2.bias is not None or c2.kernel_size != (1,):
        return None
    if not (_bn_unit_ok(features, bn1) and mbn.affine and mbn.track_running_stats and mbn.momentum is not None
            and bn2.affine and bn2.track_running_stats and bn2.momentum is not None):
        return None
    C1 = c1.weight.shape[0]
    Cla = mconv.weight.shape[0]
    if (features.shape[1] != c1.weight.shape[1] or mconv.weight.shape[1] != 3 + 2 * C1 or c2.weight.shape[1] != Cla
            or C1 % 4 or Cla % 4):
        return None
    if not _lib.lib().cl3d_fused_supported(10, int(la.nsample), int(Cla)):
        return None
    if not (features.is_cuda and query_xyz.is_cuda and query_xyz.dim() == 3 and features.shape[2] == support_xyz.shape[1]):
        return None
    sc, sbn = (shortcut[0], shortcut[1]) if shortcut is not None else (None, None)
    if sbn is not None and not sbn.training:
        return None
    if not _residual_ok(c2, identity, sc, sbn, features.device, features.shape[0], query_xyz.shape[1]):
        return None  # (checked before anything runs: a late refusal would have updated running statistics twice)
    features, query_xyz, support_xyz, query_mask, support_mask = _checked(features, query_xyz, support_xyz, query_mask, support_mask)
    prec = PRECISIONS[precision]
    params = (c1.weight, bn1.weight, bn1.bias, mconv.weight, mbn.weight, mbn.bias, c2.weight, bn2.weight, bn2.bias)
    need_grad = _wants_grad(features, identity, *params)
    idx, _ = _query(query_xyz, support_xyz, query_mask, support_mask, la.radius, la.nsample, need_grad)
    y1 = _Conv1x1.apply(features, c1.weight.view(C1, -1), prec)
    ght, wr = _BnReluPointRows.apply(y1, bn1.weight, bn1.bias, bn1, mconv.weight.view(Cla, 3 + 2 * C1), prec)
    rows, scale, shift = _PointwiseMLP.apply(ght, wr, mbn.weight, mbn.bias, mbn.running_mean, mbn.running_var,
                                             query_xyz.contiguous(), support_xyz.contiguous(), idx, la.radius, True,
                                             mbn.momentum, mbn.eps, need_grad, _step_counter(mbn), True)
    y2 = _Conv1x1Rows.apply(rows, scale, shift, c2.weight.view(c2.weight.shape[0], Cla), prec)
    return conv_bn_act(None, c2, bn2, relu=True, residual=identity,
                       res_conv=shortcut[0] if shortcut is not None else None,
                       res_bn=shortcut[1] if shortcut is not None else None, precision=precision, conv_out=y2)


class _ReduceBottleneckLA(Function):
    """The PosPool / AdaptiveWeight / PseudoGrid operator INSIDE a bottleneck, without the [B,C,N] tensors either side of
    it (backbones/resnet.py:32-39,47-66):
        y1 [B,C,N]  conv1's RAW output  ->  batch statistics of BatchNorm 1  ->  the layout change to point-major rows
                    applies max(scale1 y1 + shift1, 0) on the way (cl3d_transpose_bn_relu: conv1's activated tensor is
                    never written)  ->  the fused reduction, writing POINT-MAJOR rows [B,M,C]  ->  batch statistics of
                    the operator's own BatchNorm on the rows (cl3d_bn_rows_stats)
    returns (rows, scale2, shift2): the consumer (conv2, _Conv1x1Rows) applies max(scale2 rows + shift2, 0) while it stages
    its operand and hands back the gradient with respect to the ACTIVATED rows; backward: BatchNorm 2 + ReLU backward on
    rows (cl3d_bn_rows_bwd), the support-major pass reading those rows directly (no transposition of the upstream
    gradient), BatchNorm 1 + ReLU backward on (d act, y1) with the mask recomputed from y1 (cl3d_bn_relu_bwd)."""

    @staticmethod
    def forward(ctx, y1, gamma1, beta1, bn1, p0, p1, op, query_xyz, support_xyz, query_mask, idx, idx_mask, radius,
                normalize, reduction, pint, pfloat, constant, gamma2, beta2, bn2, need_grad=True):
        y1 = y1.contiguous()
        B, C, N = y1.shape
        _, M, K = idx.shape
        dev = y1.device
        lib = _lib.lib()
        vec1 = torch.empty((4, C), dtype=torch.float32, device=dev)
        vec2 = torch.empty((4, C), dtype=torch.float32, device=dev)
        nparts1 = lib.cl3d_bn_partials(B, C, N)
        nparts2 = lib.cl3d_bn_rows_partials(B * M, C)
        partial1 = torch.empty((nparts1, C, 2), dtype=torch.float64, device=dev)
        partial2 = torch.empty((nparts2, C, 2), dtype=torch.float64, device=dev)
        ft = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        rows = torch.empty((B, M, C), dtype=torch.float32, device=dev)
        # (a training-mode forward under torch.no_grad() -- BatchNorm recalibration, evaluation with batch statistics --
        # keeps nothing for a backward and builds no CSR table: ADVICE r5)
        slotrec = torch.empty((B, M, K, 4), dtype=torch.float32, device=dev) if need_grad else None
        pairs = None
        if need_grad and op == OP_PSEUDOGRID and C % 4 == 0 and not constant:
            pairs = torch.empty((B, M, K, 8), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            st = _stream(y1)
            _lib.check(lib.cl3d_bn_relu_stats(_p(y1), B, C, N, _p(partial1), nparts1, float(B * N), float(bn1.eps),
                                              float(bn1.momentum), _p(gamma1), _p(beta1), _p(bn1.running_mean),
                                              _p(bn1.running_var), _p(_step_counter(bn1)), _p(vec1[0]), _p(vec1[1]),
                                              _p(vec1[2]), _p(vec1[3]), st))
            _lib.check(lib.cl3d_transpose_bn_relu(_p(y1), _p(vec1[0]), _p(vec1[1]), B, C, N, _p(ft), st))
            pre = _mark(dev) if need_grad else None
            wait_ready(idx)
            _lib.check(lib.cl3d_fused_reduce_fwd(
                op, _p(query_xyz), _p(support_xyz), _p(query_mask), _p(idx), _p(idx_mask), _p(ft), B, N, M, K, C,
                float(radius), int(normalize), reduction, _p(p0), _p(p1), pint, float(pfloat), int(constant),
                _p(rows), 0, _p(slotrec), _p(pairs), st))
            if need_grad:
                _start_inverse(idx, N, pre)
            _lib.check(lib.cl3d_bn_rows_stats(_p(rows), B * M, C, _p(partial2), nparts2, float(B * M), float(bn2.eps),
                                              float(bn2.momentum), _p(gamma2), _p(beta2), _p(bn2.running_mean),
                                              _p(bn2.running_var), _p(_step_counter(bn2)), _p(vec2[0]), _p(vec2[1]),
                                              _p(vec2[2]), _p(vec2[3]), st))
        ctx.save_for_backward(y1, vec1, gamma1, ft, slotrec, p0, p1, pairs, rows, vec2, gamma2)
        ctx.idx = idx
        ctx.meta = (op, B, N, M, K, C, pint, pfloat, constant, nparts1, nparts2)
        _join_inverse(idx)  # (behind the statistics pass: the build is longer than the gather pass it runs beside)
        scale2, shift2 = vec2[0], vec2[1]
        ctx.mark_non_differentiable(scale2, shift2)
        return rows, scale2, shift2

    @staticmethod
    def backward(ctx, g_act, *unused):
        y1, vec1, gamma1, ft, slotrec, p0, p1, pairs, rows, vec2, gamma2 = ctx.saved_tensors
        op, B, N, M, K, C, pint, pfloat, constant, nparts1, nparts2 = ctx.meta
        dev = y1.device
        lib = _lib.lib()
        g_act = g_act.contiguous()
        drows = torch.empty_like(rows)
        coef2 = torch.empty((5, C), dtype=torch.float32, device=dev)
        coef1 = torch.empty((5, C), dtype=torch.float32, device=dev)
        partial2 = torch.empty((nparts2, C, 2), dtype=torch.float64, device=dev)
        partial1 = torch.empty((nparts1, C, 2), dtype=torch.float64, device=dev)
        dact = torch.empty((B, C, N), dtype=torch.float32, device=dev)
        dy1 = torch.empty_like(y1)
        nparts = lib.cl3d_fused_param_partials(op, B, N, C)
        npar = {OP_ADAPTIVE: 4, OP_PSEUDOGRID: 16}.get(op, 0)
        dparam = torch.empty((nparts, C, npar), dtype=torch.float32, device=dev) if nparts else None
        g0 = g1 = None
        if op == OP_ADAPTIVE:
            g0 = torch.empty((C // pint, 3), dtype=torch.float32, device=dev)
            g1 = torch.empty((C // pint,), dtype=torch.float32, device=dev)
        elif op == OP_PSEUDOGRID:
            g1 = torch.empty((pint, C), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            st = _stream(y1)
            _lib.check(lib.cl3d_bn_rows_bwd(_p(g_act), _p(rows), _p(vec2[0]), _p(vec2[1]), _p(vec2[2]), _p(vec2[3]), _p(gamma2),
                                            B * M, C, float(B * M), _p(partial2), nparts2, _p(coef2), _p(drows), st))
            off, slots = inverse_index(ctx.idx, N)
            _lib.check(lib.cl3d_fused_reduce_bwd(op, _p(drows), _p(ft), _p(slotrec), _p(pairs), _p(ctx.idx), _p(off), _p(slots),
                                                 B, N, M, K, C, _p(p0), _p(p1), pint, float(pfloat), int(constant), _p(dact), 1,
                                                 _p(dparam), nparts, st))
            if g1 is not None:
                _lib.check(lib.cl3d_fused_param_reduce(op, _p(dparam), nparts, C, pint, _p(g0), _p(g1), st))
            _lib.check(lib.cl3d_bn_relu_bwd(_p(dact), _p(y1), _p(vec1[0]), _p(vec1[1]), _p(vec1[2]), _p(vec1[3]), _p(gamma1),
                                            B, C, N, float(B * N), _p(partial1), nparts1, _p(coef1), _p(dy1), st))
        return (dy1, coef1[3], coef1[4], None, g0, g1) + (None,) * 12 + (coef2[3], coef2[4], None, None)


def _reduce_operator_args(la):
    """(op, p0, p1, normalize, reduction, pint, pfloat, constant) of a PosPool / AdaptiveWeight / PseudoGrid module, as
    pospool() / adaptive_weight() / pseudo_grid() hand them to the kernels; None when the fused kernels do not cover it."""
    kind = type(la).__name__
    if kind == 'PosPool' and _supported('pospool', la):
        C = la.in_channels
        if la.position_embedding == 'xyz':
            return None if C % 3 else (OP_POSPOOL_XYZ, None, None, True, _RED[la.reduction], 0, 0.0, False)
        if C % 6:
            return None
        fd = C // 6
        dev = next(la.parameters()).device
        p0 = torch.pow(1.0 * 1000, (1.0 / fd) * torch.arange(fd, dtype=torch.float32, device=dev))
        return (OP_POSPOOL_SINCOS, p0, None, True, _RED[la.reduction], 0, 0.0, False)
    if kind == 'AdaptiveWeight' and _supported('adaptive_weight', la):
        conv = la.mlps.conv0
        return (OP_ADAPTIVE, conv.weight.view(conv.weight.shape[0], 3), conv.bias, True, _RED[la.reduction],
                int(la.shared_channels), 0.0, False)
    if kind == 'PseudoGrid' and _supported('pseudo_grid', la):
        return (OP_PSEUDOGRID, la.K_points.contiguous(), la.kernel_weights, False, _RED['sum'], int(la.K_points.shape[0]),
                1.0 / float(la.extent), la.KP_influence == 'constant')
    return None


def reduce_bottleneck(conv1, la, conv2, shortcut, query_xyz, support_xyz, query_mask, support_mask, features, identity,
                      precision='f32'):
    """A whole PosPool / AdaptiveWeight / PseudoGrid bottleneck in training mode without the [B,C,N] tensors between its
    layers (the counterpart of pointwise_bottleneck for the three gather-and-reduce operators):
        y1 = conv1(x)                                    MFMA convolution, raw output
        rows, scale, shift = operator(act1(y1))          _ReduceBottleneckLA: BatchNorm + ReLU of conv1 in the layout change,
                                                         the operator's result as point-major rows, its BatchNorm as statistics
        y2 = conv2(act(rows))                            conv2 reads the rows, the operator's BatchNorm + ReLU in ITS staging
        out = ReLU(BN2(y2) + shortcut)                   the fused tail of conv_bn_act
    Returns None when the configuration is outside what the kernels cover (the caller then runs layer by layer)."""
    c1, bn1 = conv1[0], conv1[1]
    c2, bn2 = conv2[0], conv2[1]
    if getattr(la, 'output_conv', True) or not hasattr(la, 'out_transform'):
        return None  # (an operator whose output transform carries a convolution: layer by layer)
    obn = la.out_transform[0]
    if not (bn1.training and bn2.training and obn.training and la.training):
        return None
    if c1.bias is not None or c1.kernel_size != (1,) or c2.bias is not None or c2.kernel_size != (1,):
        return None
    if not (_bn_unit_ok(features, bn1) and _bn_ok(bn1) and obn.affine and obn.track_running_stats and obn.momentum is not None
            and bn2.affine and bn2.track_running_stats and bn2.momentum is not None):
        return None
    C1 = c1.weight.shape[0]
    if (features.shape[1] != c1.weight.shape[1] or la.in_channels != C1 or la.out_channels != C1 or c2.weight.shape[1] != C1
            or C1 % 4 or C1 > 1024 or obn.num_features != C1):
        return None
    if not (features.is_cuda and query_xyz.is_cuda and query_xyz.dim() == 3 and features.shape[2] == support_xyz.shape[1]):
        return None
    oargs = _reduce_operator_args(la)
    if oargs is None:
        return None
    sc, sbn = (shortcut[0], shortcut[1]) if shortcut is not None else (None, None)
    if sbn is not None and not sbn.training:
        return None
    if not _residual_ok(c2, identity, sc, sbn, features.device, features.shape[0], query_xyz.shape[1]):
        return None
    features, query_xyz, support_xyz, query_mask, support_mask = _checked(features, query_xyz, support_xyz, query_mask, support_mask)
    prec = PRECISIONS[precision]
    op, p0, p1, normalize, reduction, pint, pfloat, constant = oargs
    need_grad = _wants_grad(features, identity, c1.weight, bn1.weight, bn1.bias, p0, p1, obn.weight, obn.bias, c2.weight,
                            bn2.weight, bn2.bias)
    idx, idx_mask = _query(query_xyz, support_xyz, query_mask, support_mask, la.radius, la.nsample, need_grad)
    y1 = _Conv1x1.apply(features, c1.weight.view(C1, -1), prec)
    rows, scale, shift = _ReduceBottleneckLA.apply(y1, bn1.weight, bn1.bias, bn1, p0, p1, op, query_xyz, support_xyz,
                                                   query_mask, idx, idx_mask, la.radius, normalize, reduction, pint, pfloat,
                                                   constant, obn.weight, obn.bias, obn, need_grad)
    y2 = _Conv1x1Rows.apply(rows, scale, shift, c2.weight.view(c2.weight.shape[0], C1), prec)
    return conv_bn_act(None, c2, bn2, relu=True, residual=identity,
                       res_conv=shortcut[0] if shortcut is not None else None,
                       res_bn=shortcut[1] if shortcut is not None else None, precision=precision, conv_out=y2)


def point_rows(features, W, precision='f32'):
    """[G | H] rows and W_r of the factored PointWiseMLP contraction; precision 'f32' or 'bf16' (inputs of the
    contraction rounded to bf16, f32 accumulation; coordinates, indices and BatchNorm statistics stay f32)."""
    return _PointRows.apply(features, W, PRECISIONS[precision])


# the one-call-per-pass path is taken outside HIP-graph capture, for a stand-alone operator (no per-forward ball-query
# memo: a backbone that shares geometry between its blocks keeps the kernel-by-kernel path, which shares it)
PASS_CALLS = True


MAXPOOL_TARGETS = True  # max pooling keeps support indices and scatters its gradient (see _MaxPool); False: slot bytes + CSR gather

PW_CSR_FIRST = True  # (module attribute: scripts set it to False for the A/B; see pointwise_mlp)
# the same order for a STAND-ALONE PosPool / AdaptiveWeight / PseudoGrid step (no per-forward geometry memo): 0.269 -> 0.264,
# 0.268 -> 0.2685, 0.393 -> 0.386 ms, two alternating pairs; inside a backbone the build stays behind the consumer (_query)
REDUCE_CSR_FIRST = True


def _use_pass_calls(training, need_grad, bn):
    return (PASS_CALLS and training and need_grad and pt_utils._BQ_CACHE is None and pt_utils.ASYNC_INDEX == 'auto'
            and not torch.cuda.is_current_stream_capturing() and bn.running_mean is not None and bn.track_running_stats)


def pointwise_mlp(query_xyz, support_xyz, query_mask, support_mask, features, radius, nsample, mlps, reduction,
                  training, precision='f32'):
    assert reduction == 'max'
    features, query_xyz, support_xyz, query_mask, support_mask = _checked(features, query_xyz, support_xyz, query_mask, support_mask)
    conv, bn = mlps.conv0[0], mlps.conv0[1]
    need_grad = training and _wants_grad(features, conv.weight, bn.weight, bn.bias)
    C = features.shape[1]
    Co = conv.weight.shape[0]
    W = conv.weight.view(Co, 3 + 2 * C)
    if _use_pass_calls(training, need_grad, bn):
        from .pass_calls import _PointwiseMLPPass  # (one C-ABI call per pass: pass_calls.py)
        return _PointwiseMLPPass.apply(features, W.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                       _step_counter(bn), query_xyz, support_xyz, query_mask, support_mask, radius,
                                       nsample, bn.momentum, bn.eps, PRECISIONS[precision], True)
    # Capture order (round 5, second session; profiles/r05/pw_capture_order.txt): query, per-point product, CSR build, and
    # only then the gather pass (inside _PointwiseMLP.forward).  The runtime lays a captured step out depth-first along each
    # node's FIRST-captured dependent (_start_inverse): with the build captured before the gather pass, the build inherits
    # the query's queue and its count pass runs the moment the query ends -- 11 us, alone, instead of 60 us squeezed in
    # behind the gather pass's workgroups -- while the gather pass joins the product's queue (it pays one cross-queue
    # hand-over either way: behind the product before, behind the query now).  The rest of the build then finishes under the
    # gather pass instead of beside bwd_rows: TRAIN 71 -> 62 us, bwd_rows 34 -> 23, step 0.2905 -> 0.2841 ms (three
    # alternating pairs).  PW_CSR_FIRST = False restores the order of the round's first session (build behind the consumer),
    # which stays the better one for the gather-and-reduce operators inside backbones (_query's note).
    if PW_CSR_FIRST and need_grad:
        idx, _ = _query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample, need_grad)
        ght, wr = point_rows(features, W, precision)
        _start_inverse(idx, support_xyz.shape[1], None)
    else:
        idx, _ = _query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample, need_grad)
        ght, wr = point_rows(features, W, precision)
    use_batch_stats = training or bn.running_mean is None
    momentum = bn.momentum  # never None here: use_fused() sends that configuration to the grouped path
    return _PointwiseMLP.apply(ght.contiguous(), wr, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                               query_xyz.contiguous(), support_xyz.contiguous(), idx, radius, use_batch_stats,
                               momentum, bn.eps, _wants_grad(features, conv.weight, bn.weight, bn.bias),
                               _step_counter(bn) if use_batch_stats and bn.track_running_stats else None)


# ---------------------------------------------------------------- the 1x1 convolutions and BatchNorm tails of a bottleneck
class _Conv1x1(Function):
    """y [B,Co,N] = W [Co,C] x [B,C,N] on the matrix cores (csrc/mfma_gemm.hip), with both gradients; replaces the
    library's Conv1d forward / backward-data / backward-weight kernels of backbones/resnet.py:32-39,58-66."""

    @staticmethod
    def forward(ctx, x, W, precision, residual=None):
        """residual [B,Co,N] (optional) is added in the product's epilogue: y = W x + residual."""
        x = x.contiguous()
        W = W.contiguous()
        B, C, N = x.shape
        Co = W.shape[0]
        y = torch.empty((B, Co, N), dtype=torch.float32, device=x.device)
        ws, ws_bytes = _gemm_scratch(15, B, N, Co, C, x.device)
        with _lib.on_device(x.device):
            if residual is None:
                _lib.check(_lib.lib().cl3d_conv1x1_fwd(_p(x), _p(W), B, C, N, Co, precision, _p(y), _p(ws), ws_bytes, _stream(x)))
            else:
                residual = residual.contiguous()
                _lib.check(_lib.lib().cl3d_conv1x1_bn_act_fwd(_p(x), _p(W), None, None, _p(residual), 0, B, C, N, Co, precision,
                                                              _p(y), _p(ws), ws_bytes, _stream(x)))
        ctx.save_for_backward(x, W)
        ctx.precision = precision
        ctx.has_residual = residual is not None
        ctx.w_leaf = _weight_leaf(W)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        B, C, N = x.shape
        Co = W.shape[0]
        dy = dy.contiguous()
        lib = _lib.lib()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(W) if ctx.needs_input_grad[1] else None
        prec = ctx.precision

        def data_grad():
            if dx is not None:
                ws, ws_bytes = _gemm_scratch(15, B, N, Co, C, x.device)
                _lib.check(lib.cl3d_conv1x1_bwd_data(_p(dy), _p(W), B, C, N, Co, prec, _p(dx), _p(ws), ws_bytes, _stream(x)))

        def weight_grad():
            if dW is not None:
                ws, ws_bytes = _gemm_scratch(15, B, N, Co, C, x.device)
                _lib.check(lib.cl3d_conv1x1_bwd_weight(_p(x), _p(dy), B, C, N, Co, prec, _p(dW), _p(ws), ws_bytes, _stream(x)))

        with _lib.on_device(x.device):
            if _fork_join(x.device, weight_grad, data_grad, B * N, (ctx.w_leaf, dW, (x, dy))):
                dW = None  # handed to the parameter by join_weight_gradients()
        return dx, dW, None, (dy if ctx.has_residual and ctx.needs_input_grad[3] else None)


def decode_level(up, fine_xyz, coarse_xyz, fine_mask, coarse_mask, coarse_feats, skip_feats, conv, bn, precision='f32'):
    """One level of the segmentation decoders (heads/segmentation_head.py:55-73): nearest up-sampling of the coarse
    features, concatenation with the skip features, 1x1 Conv1d + BatchNorm1d + ReLU -- without the concatenated tensor.
    Nearest up-sampling is a column gather and commutes with a 1x1 convolution, and the convolution splits over the two
    channel groups:  W [up(f) ; s] = up(W_a f) + W_b s.  So W_a runs at the COARSE resolution (a quarter of the points
    or fewer), the gather moves C_out channels instead of C_up (four times fewer at the first level), and the sum is the
    epilogue of the skip branch's product.  Same weight tensor as the reference's (its two column blocks); each output
    element is the same dot product, added in two parts.  Returns None when the modules are outside what the kernels
    cover (the caller then concatenates as the reference does)."""
    if getattr(up, 'mode', None) != 'nearest':  # only a plain column gather commutes with the convolution
        return None
    if conv.bias is not None or conv.kernel_size != (1,) or not _bn_unit_ok(skip_feats, bn) or not coarse_feats.is_cuda:
        return None
    if coarse_feats.dtype != torch.float32 or coarse_feats.dim() != 3:
        return None
    if (skip_feats.shape[0] != coarse_feats.shape[0] or skip_feats.shape[2] != fine_xyz.shape[1]
            or coarse_feats.shape[2] != coarse_xyz.shape[1] or skip_feats.device != coarse_feats.device
            or conv.weight.shape[1] != coarse_feats.shape[1] + skip_feats.shape[1]):
        return None  # shapes the raw-pointer kernels would read out of bounds on: the modules raise the proper error
    training = bn.training
    if not training and _wants_grad(coarse_feats, skip_feats, conv.weight, bn.weight, bn.bias):
        return None  # backward through frozen statistics: the nn modules do it
    prec = PRECISIONS[precision]
    Cu, Cs = coarse_feats.shape[1], skip_feats.shape[1]
    Co = conv.weight.shape[0]
    if conv.weight.shape[1] != Cu + Cs:
        return None
    W = conv.weight.view(Co, Cu + Cs)
    a = _Conv1x1.apply(coarse_feats, W[:, :Cu], prec)
    g = up(fine_xyz, coarse_xyz, fine_mask, coarse_mask, a)
    y = _Conv1x1.apply(skip_feats, W[:, Cu:], prec, g)
    if training:
        return _BnAddRelu.apply(y, bn.weight, bn.bias, None, None, None, bn, None, True)
    return bn_relu(y, bn)


class _BnAddRelu(Function):
    """out = ReLU(BN1(x1) + R), R = 0 | x2 | BN2(x2), training-mode statistics (csrc/bn_relu.hip).  One launch each way
    when a channel has few values (deep stages), statistics + apply passes otherwise; x1, x2 and the output are what is
    kept for the backward pass."""

    @staticmethod
    def forward(ctx, x1, gamma1, beta1, x2, gamma2, beta2, bn1, bn2, relu):
        x1 = x1.contiguous()
        B, C, N = x1.shape
        dev = x1.device
        lib = _lib.lib()
        out = torch.empty_like(x1)
        vec1 = torch.empty((4, C), dtype=torch.float32, device=dev)
        vec2 = torch.empty((4, C), dtype=torch.float32, device=dev) if bn2 is not None else None
        if x2 is not None:
            x2 = x2.contiguous()
        nparts = lib.cl3d_bn_partials(B, C, N)
        partial = torch.empty((nparts, C, 2), dtype=torch.float64, device=dev)
        with _lib.on_device(dev):
            _lib.check(lib.cl3d_bn_add_relu_train_fwd(
                _p(x1), _p(gamma1), _p(beta1), _p(bn1.running_mean), _p(bn1.running_var), _p(_step_counter(bn1)),
                float(bn1.eps), float(bn1.momentum),
                _p(x2), _p(gamma2) if bn2 is not None else None, _p(beta2) if bn2 is not None else None,
                _p(bn2.running_mean) if bn2 is not None else None, _p(bn2.running_var) if bn2 is not None else None,
                _p(_step_counter(bn2)) if bn2 is not None else None,
                float(bn2.eps) if bn2 is not None else 0.0, float(bn2.momentum) if bn2 is not None else 0.0, int(relu),
                B, C, N, _p(partial), nparts, _p(vec1), _p(vec2), _p(out), _stream(x1)))
        ctx.save_for_backward(x1, x2, out, vec1, vec2, gamma1, gamma2)
        ctx.relu = relu
        return out

    @staticmethod
    def backward(ctx, g):
        x1, x2, out, vec1, vec2, gamma1, gamma2 = ctx.saved_tensors
        B, C, N = x1.shape
        dev = x1.device
        g = g.contiguous()
        lib = _lib.lib()
        nparts = lib.cl3d_bn_partials(B, C, N)
        partial = torch.empty((2, nparts, C, 2), dtype=torch.float64, device=dev)
        coef1 = torch.empty((5, C), dtype=torch.float32, device=dev)
        coef2 = torch.empty((5, C), dtype=torch.float32, device=dev) if vec2 is not None else None
        dx1 = torch.empty_like(x1)
        dx2 = torch.empty_like(x2) if x2 is not None else None
        with _lib.on_device(dev):
            _lib.check(lib.cl3d_bn_add_relu_bwd(
                _p(g), _p(out), _p(x1), _p(vec1[2]), _p(vec1[3]), _p(gamma1), _p(x2),
                _p(vec2[2]) if vec2 is not None else None, _p(vec2[3]) if vec2 is not None else None,
                _p(gamma2) if vec2 is not None else None, int(ctx.relu), B, C, N, float(B * N), _p(partial), nparts,
                _p(coef1), _p(coef2), _p(dx1), _p(dx2), _stream(g)))
        return (dx1, coef1[3], coef1[4], dx2, coef2[3] if coef2 is not None else None,
                coef2[4] if coef2 is not None else None, None, None, None)


def _bn_unit_ok(x, bn):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and bn.affine and bn.track_running_stats
            and bn.momentum is not None)


def _residual_ok(conv, residual, res_conv, res_bn, device, B, N):
    """The kernels take the residual as a raw pointer: its dtype, device, rank and shape are checked here, against the
    tensor it is added to ([B,Co,N]) or the shortcut convolution's input ([B,Cr,N])."""
    if residual is None:
        return res_conv is None
    if res_conv is not None:
        if (res_conv.bias is not None or res_conv.kernel_size != (1,) or not _bn_unit_ok(residual, res_bn)
                or residual.shape[1] != res_conv.weight.shape[1] or res_conv.weight.shape[0] != conv.weight.shape[0]):
            return False
    elif not (residual.is_cuda and residual.dtype == torch.float32 and residual.dim() == 3
              and residual.shape[1] == conv.weight.shape[0]):
        return False
    return residual.device == device and residual.shape[0] == B and residual.shape[2] == N


def _folded(bn):
    invstd = torch.rsqrt(bn.running_var.double() + bn.eps)
    scale = bn.weight.double() * invstd
    return scale.float(), (bn.bias.double() - bn.running_mean.double() * scale).float()


def conv_bn_act(x, conv, bn, relu=True, residual=None, res_conv=None, res_bn=None, precision='f32', conv_out=None):
    """act(BN(conv(x)) + R): a bottleneck's conv1 (R = 0) or its tail conv2 + shortcut + add + ReLU
    (backbones/resnet.py:32-39,58-66), on the engine: MFMA 1x1 convolutions, one statistics pass per BatchNorm and one
    fused apply / add / ReLU pass.  Inference (no gradient): BatchNorm folded into the convolution's epilogue, one
    launch per convolution.  Returns None when the modules are outside what the kernels cover."""
    if conv_out is not None:
        # training-mode tail on a convolution the caller has already run (pointwise_bottleneck: conv2 fed by the
        # operator's rows): x is not needed, everything below the convolution is what this function does anyway
        if not bn.training or not _bn_unit_ok(conv_out, bn):
            raise RuntimeError("conv_bn_act(conv_out=...) is the training-mode tail only")
        x = conv_out
    elif conv.bias is not None or conv.kernel_size != (1,) or not _bn_unit_ok(x, bn) or x.shape[1] != conv.weight.shape[1]:
        return None
    if not _residual_ok(conv, residual, res_conv, res_bn, x.device, x.shape[0], x.shape[2]):
        return None
    prec = PRECISIONS[precision]
    Co, C = conv.weight.shape[0], conv.weight.shape[1]
    W = conv.weight.view(Co, C)
    training = bn.training
    params = [conv.weight, bn.weight, bn.bias] + ([res_conv.weight, res_bn.weight, res_bn.bias] if res_conv is not None else [])
    if not training:
        if _wants_grad(x, residual, *params):
            return None  # backward through frozen statistics: the nn modules do it
        x = x.contiguous()
        B, _, N = x.shape
        lib = _lib.lib()
        with _lib.on_device(x.device):
            st = _stream(x)
            res = residual.contiguous() if residual is not None else None
            if res_conv is not None:
                s2, t2 = _folded(res_bn)
                Cr = res_conv.weight.shape[1]
                ys = torch.empty((B, Co, N), dtype=torch.float32, device=x.device)
                ws, ws_bytes = _gemm_scratch(15, B, N, Co, Cr, x.device)
                _lib.check(lib.cl3d_conv1x1_bn_act_fwd(_p(res), _p(res_conv.weight.view(Co, Cr).contiguous()), _p(s2), _p(t2),
                                                       None, 0, B, Cr, N, Co, prec, _p(ys), _p(ws), ws_bytes, st))
                res = ys
            s1, t1 = _folded(bn)
            y = torch.empty((B, Co, N), dtype=torch.float32, device=x.device)
            ws, ws_bytes = _gemm_scratch(15, B, N, Co, C, x.device)
            _lib.check(lib.cl3d_conv1x1_bn_act_fwd(_p(x), _p(W.contiguous()), _p(s1), _p(t1), _p(res), int(relu), B, C, N, Co,
                                                   prec, _p(y), _p(ws), ws_bytes, st))
        return y
    if conv_out is not None:
        y1 = conv_out
        x2 = residual
        if res_conv is not None:
            x2 = _Conv1x1.apply(residual, res_conv.weight.view(Co, res_conv.weight.shape[1]), prec)
    else:
        y1 = _Conv1x1.apply(x, W, prec)
        x2 = residual
        if res_conv is not None:
            x2 = _Conv1x1.apply(residual, res_conv.weight.view(Co, res_conv.weight.shape[1]), prec)
    return _BnAddRelu.apply(y1, bn.weight, bn.bias, x2, res_bn.weight if res_bn is not None else None,
                            res_bn.bias if res_bn is not None else None, bn, res_bn, relu)

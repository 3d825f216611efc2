"""Grouping API of the reference (`pytorch/ops/pt_custom_ops/pt_utils.py`) on the HIP engine.

Same public names, constructor arguments, forward signatures and return tuples as the reference
module, so `models/local_aggregation_operators.py`, `models/backbones/resnet.py` and
`models/heads/segmentation_head.py` of the reference run unchanged when this file is what
`from pt_utils import ...` resolves to (see INTEGRATION.md and `drop_in/`).

Differences that are implementation, not behaviour:
  * `MaskedQueryAndGroup` issues one fused C-ABI call (`cl3d_group_xyz_features`) for what the
    reference does with a transpose, two generic gathers and two in-place element-wise passes
    (pt_utils.py:124-132);
  * an optional per-forward ball-query memo (`ball_query_cache`) removes the identical searches the
    reference repeats inside every strided bottleneck.
"""
import contextlib
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _ext


# --------------------------------------------------------------------------- autograd wrappers
class GroupingOperation(Function):
    """features (B,C,N), idx (B,M,K) int32 -> (B,C,M,K).  Reference: pt_utils.py:16-61."""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n_support = features.size(2)
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.group_points_grad(grad_out.contiguous(), idx, ctx.n_support), None


grouping_operation = GroupingOperation.apply


class MaskedOrderedBallQuery(Function):
    """Reference: pt_utils.py:67-77.  Outputs are index tensors, non-differentiable."""

    @staticmethod
    def forward(ctx, radius, nsample, query_xyz, support_xyz, query_mask, support_mask):
        inds, inds_mask = _ball_query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample)
        ctx.mark_non_differentiable(inds, inds_mask)
        return inds, inds_mask

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None, None


masked_ordered_ball_query = MaskedOrderedBallQuery.apply


class MaskedNearestQuery(Function):
    """Reference: pt_utils.py:83-92."""

    @staticmethod
    def forward(ctx, query_xyz, support_xyz, query_mask, support_mask):
        inds, inds_mask = _ext.masked_nearest_query(query_xyz, support_xyz, query_mask, support_mask)
        ctx.mark_non_differentiable(inds, inds_mask)
        return inds, inds_mask

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None


masked_nearest_query = MaskedNearestQuery.apply


class MaskedGridSubsampling(Function):
    """Reference: pt_utils.py:98-108."""

    @staticmethod
    def forward(ctx, xyz, mask, npoint, sampleDl):
        sub_xyz, sub_mask = _ext.masked_grid_subsampling(xyz, mask, npoint, sampleDl)
        ctx.mark_non_differentiable(sub_xyz, sub_mask)
        return sub_xyz, sub_mask

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None


masked_grid_subsampling = MaskedGridSubsampling.apply


# ----------------------------------------------------------------- per-forward ball-query memo
_BQ_CACHE = None


@contextlib.contextmanager
def ball_query_cache():
    """Within the context, ball queries with identical (tensors, radius, nsample) are computed once.

    Keys hold the tensors' storage pointers *and versions*, and the cache keeps the key tensors
    alive, so a recycled allocation can never alias a stale entry.  Meant to wrap one model forward.
    """
    global _BQ_CACHE
    prev, _BQ_CACHE = _BQ_CACHE, {}
    try:
        yield
    finally:
        _BQ_CACHE = prev


# Index structures (ball query, CSR inverse) depend on coordinates only, so they can be built on side HIP
# streams while the main stream goes on with feature work (layout changes, per-point GEMMs, the forward
# kernels that do not need the CSR); consumers wait on an event right before the first kernel that reads
# them.  ASYNC_INDEX: 'auto' = do so while a HIP graph is being captured (the forks become parallel branches
# of the graph; measured -6 % on a PointWiseMLP step) and stay on the caller's stream in eager mode, where
# the host, not the GPU, sets the pace and the extra stream/event calls cost more than the overlap returns;
# True / False force it (CL3D_ASYNC=1 / 0).
ASYNC_INDEX = {'1': True, '0': False}.get(os.environ.get('CL3D_ASYNC', ''), 'auto')


def async_index():
    if ASYNC_INDEX == 'auto':
        return torch.cuda.is_current_stream_capturing()
    return bool(ASYNC_INDEX)


_INDEX_STREAMS = {}


def index_stream(device, which=0):
    """which=0: ball queries (and the prefetched pyramid); which=1: CSR builds (each joined through its own event)."""
    st = _INDEX_STREAMS.get((device, which))
    if st is None:
        # (a high-priority ball-query stream -- the query is on the step's critical path, the per-point GEMM beside it
        # has slack -- was measured in round 3: no effect on the replayed step, 0.368-0.373 ms either way)
        st = _INDEX_STREAMS[(device, which)] = torch.cuda.Stream(device=device)
    return st


def wait_ready(t):
    """Make the current stream wait for a tensor produced on the index stream (no-op otherwise)."""
    ev = getattr(t, '_cl3d_ready', None)
    if ev is not None:
        torch.cuda.current_stream(t.device).wait_event(ev)


_CONSUMER_STREAM = None  # set by prefetch_geometry: the compute stream that will read what is produced ahead of it


# ---- which ball-query implementation, by measurement.  For clouds of up to 4096 points two paths
# apply -- `tile` (the cloud resident in one CU's LDS) and `cells` (cell grid through HBM scratch) -- and which is faster
# depends on how many points fall inside the radius (the reference's ">3K candidates" regime,
# masked_ordered_ball_query_gpu.cu:58-75): at the metric shape tile wins at a mean in-radius count of 1.5 K (53 against
# 66 us) and loses at 4 K (260 against 203 us).  The density is a property of the data, the geometry of a stage is static
# per (batch shape, radius): the first time a key is seen OUTSIDE a stream capture both paths are timed (two extra
# launches each, once, with a host wait -- off the steady state) and the winner is kept; inside a capture, or with the
# tuner off, an unseen key takes the library's choice.
BQ_TUNE = True              # False: always the library's own choice by size (tests pin paths through CL3D_BQ_PATH instead)
_BQ_PATH_TABLE = {}         # key -> (path, {path: microseconds})


def bq_tune_key(device_index, B, M, N, nsample, radius):
    """Sizes, and the radius in quarter-octave buckets: a stage's radius is a constant of the configuration, so equal
    keys mean the same stage on same-shaped batches; different densities (another dataset, another radius) get their own."""
    import math
    bucket = int(round(4.0 * math.log2(radius))) if radius > 0 and math.isfinite(radius) else None
    return (int(device_index), int(B), int(M), int(N), int(nsample), bucket)


def _bq_path(query_xyz, support_xyz, query_mask, support_mask, radius, nsample):
    """0 (library's choice) or the measured winner among the applicable paths for this key."""
    if not (BQ_TUNE and query_xyz.is_cuda) or os.environ.get("CL3D_BQ_PATH"):
        return 0
    B, M, _ = query_xyz.shape
    N = support_xyz.shape[1]
    if B * M < 4096:
        return 0  # (a few microseconds either way)
    key = bq_tune_key(query_xyz.device.index or 0, B, M, N, nsample, radius)
    hit = _BQ_PATH_TABLE.get(key)
    if hit is not None:
        return hit[0]
    from . import _lib
    mask = _lib.lib().cl3d_ball_query_paths(M, N, int(nsample))
    cands = [p for p in (1, 2) if mask >> p & 1]
    if len(cands) < 2 or torch.cuda.is_current_stream_capturing():
        if len(cands) < 2:
            _BQ_PATH_TABLE[key] = (0, {})
        return 0
    times = {}
    for p in cands:
        _ext.masked_ordered_ball_query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample, path=p)  # warm
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _ext.masked_ordered_ball_query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample, path=p)
        e1.record()
        e1.synchronize()
        times[p] = e0.elapsed_time(e1) * 1e3
    best = min(times, key=times.get)
    _BQ_PATH_TABLE[key] = (best, times)
    return best


def _run_ball_query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample):
    path = _bq_path(query_xyz, support_xyz, query_mask, support_mask, radius, nsample)
    if not (query_xyz.is_cuda and async_index()):
        return _ext.masked_ordered_ball_query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample, path)
    dev = query_xyz.device
    main, side = torch.cuda.current_stream(dev), index_stream(dev)
    # inside prefetch_geometry the caller already IS on the index stream; the stream that will consume the result is
    # the one that entered the prefetch
    consumer = _CONSUMER_STREAM if (main == side and _CONSUMER_STREAM is not None) else main
    if main != side:
        side.wait_stream(main)  # the coordinates were produced on the caller's stream
    with torch.cuda.stream(side):
        out = _ext.masked_ordered_ball_query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample, path)
        ev = torch.cuda.Event()
        ev.record(side)
    capturing = torch.cuda.is_current_stream_capturing()
    if not capturing:  # a capture's private pool never hands a block to another stream mid-graph
        for t in out:
            t.record_stream(consumer)  # produced on the index stream, read on the consumer's
        for t in (query_xyz, support_xyz, query_mask, support_mask):
            t.record_stream(side)      # (possibly temporaries of .contiguous()) read on the index stream
    for t in out:
        t._cl3d_ready = ev
    return out


def _ball_query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample, defer=False):
    """(idx, idx_mask); with defer=True the caller promises to wait_ready() them before use."""
    if _BQ_CACHE is None:
        out = _run_ball_query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample)
    else:
        tensors = (query_xyz, support_xyz, query_mask, support_mask)
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors) + (float(radius), int(nsample))
        hit = _BQ_CACHE.get(key)
        if hit is None:
            out = _run_ball_query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample)
            hit = (out, tensors)
            _BQ_CACHE[key] = hit
        out = hit[0]
    if not defer:
        wait_ready(out[0])
    return out


# 'auto': prefetch when there are fewer than 8 clouds per GPU (see prefetch_geometry).
# Same-box A/B of whole backbone steps, off -> on: one 40 960-point scene 7.64 -> 7.34 ms, 4 x 10 000 points
# 7.19 -> 6.97 ms, one 81 920-point scene (width 288) 20.36 -> 20.06 ms; 16 x 4096 points 8.31 -> 8.52 ms (with 16
# clouds the one-workgroup-per-cloud kernels already fill 16 CUs and the early, heaviest layers lose more to the
# contention than the late ones gain).
PREFETCH_GEOMETRY = 'auto'  # True / False force it (tests, scripts)


def _subsample(xyz, mask, npoint, sampleDl):
    """masked_grid_subsampling through the per-forward memo (a prefetched pyramid level is picked up here)."""
    if _BQ_CACHE is None:
        return masked_grid_subsampling(xyz, mask, npoint, sampleDl)
    key = ('sub', xyz.data_ptr(), xyz._version, tuple(xyz.shape), mask.data_ptr(), mask._version, int(npoint),
           float(sampleDl))
    hit = _BQ_CACHE.get(key)
    if hit is None:
        hit = (masked_grid_subsampling(xyz, mask, npoint, sampleDl), (xyz, mask))
        _BQ_CACHE[key] = hit
    wait_ready(hit[0][0])
    return hit[0]


def prefetch_geometry(xyz, mask, radius, sampleDl, nsamples, npoints, self_queries=True):
    """Everything a strided 5-stage backbone will ask for that depends on coordinates only -- the four subsampled
    clouds, the ball query of every stage onto itself and of every strided stage onto its parent -- built on the
    index stream ahead of the feature pass, in the order the backbone consumes them.  Subsampling and the
    ball-query preparation are one-workgroup-per-cloud kernels: beside the feature kernels they cost nothing,
    in line they leave the chip idle (a 40 960-point scene: ~0.7 ms of a 7.7 ms step).  The products land in the
    `ball_query_cache()` memo, where `MaskedMaxPool` / `LocalAggregation` find them and wait on their events.
    No-op without the memo or when the index streams are off (eager mode by default, see ASYNC_INDEX)."""
    if _BQ_CACHE is None or not (xyz.is_cuda and async_index()):
        return
    # with 8 or more clouds only the sub-sampling chain goes ahead (round 3): its four launches are one workgroup per
    # cloud (16 CUs busy, 72 us each at 16 x 4096 points) and sit on the critical path when run in line, while beside
    # the first stage's feature kernels they cost nothing; the ball queries fill the chip and stay in line there
    queries = PREFETCH_GEOMETRY if PREFETCH_GEOMETRY != 'auto' else xyz.shape[0] < 8
    if not queries and PREFETCH_GEOMETRY is False:
        return
    dev = xyz.device
    main, side = torch.cuda.current_stream(dev), index_stream(dev)
    side.wait_stream(main)  # the input coordinates were produced on the caller's stream
    capturing = torch.cuda.is_current_stream_capturing()
    xyz, mask = xyz.contiguous(), mask.contiguous()
    global _CONSUMER_STREAM
    _CONSUMER_STREAM = main  # the ball queries below run with the index stream current: their outputs are read on `main`
    try:
        _prefetch_on(side, main, capturing, xyz, mask, radius, sampleDl, nsamples, npoints, self_queries, queries)
    finally:
        _CONSUMER_STREAM = None


def _prefetch_on(side, main, capturing, xyz, mask, radius, sampleDl, nsamples, npoints, self_queries, queries=True):
    with torch.cuda.stream(side):
        def query(q, s, qm, sm, r, k):
            if queries:
                _ball_query(q, s, qm, sm, r, k, defer=True)

        query(xyz, xyz, mask, mask, radius, nsamples[0])
        for stage in range(4):
            sampleDl *= 2
            sub_xyz, sub_mask = _subsample(xyz, mask, npoints[stage], sampleDl)
            ev = torch.cuda.Event()
            ev.record(side)
            for t in (sub_xyz, sub_mask):
                if not capturing:
                    t.record_stream(main)
                t._cl3d_ready = ev
            query(sub_xyz, xyz, sub_mask, mask, radius, nsamples[stage])
            radius *= 2
            if self_queries:
                query(sub_xyz, sub_xyz, sub_mask, sub_mask, radius, nsamples[stage + 1])
            xyz, mask = sub_xyz, sub_mask


def join_index_stream(device):
    """The caller's stream picks up whatever is still running on the ball-query stream (end of a forward pass
    that prefetched: a capture must not end with an unjoined branch)."""
    if device.type == 'cuda' and async_index() and (device, 0) in _INDEX_STREAMS:
        torch.cuda.current_stream(device).wait_stream(_INDEX_STREAMS[(device, 0)])


class _GroupXyzFeatures(Function):
    """(rel, grouped) of MaskedQueryAndGroup in one engine call; gradient flows to `features` only
    (coordinates are data in every caller of the reference; they never require grad there)."""

    @staticmethod
    def forward(ctx, query_xyz, support_xyz, features, idx, radius, normalize_xyz):
        rel, grouped = _ext.group_xyz_features(query_xyz, support_xyz, features, idx, radius, normalize_xyz)
        ctx.save_for_backward(idx)
        ctx.n_support = support_xyz.size(1)
        ctx.mark_non_differentiable(rel)
        return rel, grouped

    @staticmethod
    def backward(ctx, grad_rel, grad_grouped):
        (idx,) = ctx.saved_tensors
        g = _ext.group_points_grad(grad_grouped.contiguous(), idx, ctx.n_support)
        return None, None, g, None, None, None


def _group(query_xyz, support_xyz, features, idx, radius, normalize_xyz):
    if features is None:
        rel, _ = _ext.group_xyz_features(query_xyz, support_xyz, None, idx, radius, normalize_xyz)
        return rel, None
    return _GroupXyzFeatures.apply(query_xyz, support_xyz, features.contiguous(), idx, radius, normalize_xyz)


def _assemble(rel, grouped, use_xyz, ret_grouped_xyz, idx_mask):
    if grouped is None:
        assert use_xyz, "Cannot have not features and not use xyz as a feature!"
        new_features = rel
    elif use_xyz:
        new_features = torch.cat([rel, grouped], dim=1)
    else:
        new_features = grouped
    if ret_grouped_xyz:
        return new_features, rel, idx_mask
    return new_features, idx_mask


# ------------------------------------------------------------------------------------- modules
class MaskedQueryAndGroup(nn.Module):
    """Ball query + grouping.  Reference: pt_utils.py:114-144."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz

    def forward(self, query_xyz, support_xyz, query_mask, support_mask, features=None):
        idx, idx_mask = masked_ordered_ball_query(self.radius, self.nsample, query_xyz, support_xyz,
                                                  query_mask, support_mask)
        rel, grouped = _group(query_xyz, support_xyz, features, idx, self.radius, self.normalize_xyz)
        return _assemble(rel, grouped, self.use_xyz, self.ret_grouped_xyz, idx_mask)


class MaskedNearestQueryAndGroup(nn.Module):
    """1-NN query + grouping.  Reference: pt_utils.py:147-176 (its normalize_xyz=True branch reads an
    attribute that does not exist there and is never enabled; rejected here at construction)."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False):
        super().__init__()
        if normalize_xyz:
            raise AttributeError("'MaskedNearestQueryAndGroup' object has no attribute 'radius'")
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz

    def forward(self, query_xyz, support_xyz, query_mask, support_mask, features=None):
        idx, idx_mask = masked_nearest_query(query_xyz, support_xyz, query_mask, support_mask)
        rel, grouped = _group(query_xyz, support_xyz, features, idx, 1.0, False)
        return _assemble(rel, grouped, self.use_xyz, self.ret_grouped_xyz, idx_mask)


class MaskedMaxPool(nn.Module):
    """Grid subsample, then max over each sub-point's ball.  Reference: pt_utils.py:179-202."""

    def __init__(self, npoint, radius, nsample, sampleDl):
        super().__init__()
        self.npoint = npoint
        self.radius = radius
        self.nsample = nsample
        self.sampleDl = sampleDl
        self.fused = True  # set False to run the reference's gather + max_pool2d dataflow
        self.grouper = MaskedQueryAndGroup(radius, nsample, use_xyz=False, ret_grouped_xyz=True)

    def _fused_covers(self, channels):
        from . import fused
        return fused.kernels_cover('max_pool', self.nsample, channels)

    def forward(self, xyz, mask, features):
        sub_xyz, sub_mask = _subsample(xyz, mask, self.npoint, self.sampleDl)
        sub_xyz = sub_xyz.contiguous()
        sub_mask = sub_mask.contiguous()
        if self.fused and features.is_cuda and self._fused_covers(features.shape[1]):
            # one kernel, no [B,C,npoint,K] tensor (same values; same first-maximum gradient routing)
            from . import fused
            return sub_xyz, sub_mask, fused.max_pool(sub_xyz, xyz, sub_mask, mask, features, self.radius, self.nsample)
        neighborhood_features, _, _ = self.grouper(sub_xyz, xyz, sub_mask, mask, features)
        # max_pool2d, not amax: on ties its backward routes the gradient to the first maximum, as the
        # reference does; amax would split it evenly (ties are common: ReLU zeros, wrap-around padding)
        sub_features = F.max_pool2d(neighborhood_features, kernel_size=[1, neighborhood_features.shape[3]]).squeeze(-1)
        return sub_xyz, sub_mask, sub_features


class MaskedUpsample(nn.Module):
    """Feature up-sampling to a denser level.  Reference: pt_utils.py:205-227."""

    def __init__(self, radius, nsample, mode='nearest'):
        super().__init__()
        self.radius = radius
        self.nsample = nsample
        self.mode = mode
        if mode == 'nearest':
            self.grouper = MaskedNearestQueryAndGroup(use_xyz=False, ret_grouped_xyz=True)
        else:
            self.grouper = MaskedQueryAndGroup(radius, nsample, use_xyz=False, ret_grouped_xyz=True)

    def forward(self, up_xyz, xyz, up_mask, mask, features):
        neighborhood_features, _, _ = self.grouper(up_xyz, xyz, up_mask, mask, features)
        if self.mode == 'nearest':
            return neighborhood_features[..., 0].contiguous()
        if self.mode == 'max':
            return F.max_pool2d(neighborhood_features, kernel_size=[1, neighborhood_features.shape[3]]).squeeze(-1)
        raise NotImplementedError(f"mode:{self.mode} not supported in MaskedUpsample")

"""torch-CPU fp32 restatement of the reference's Python hot path -- TEST INFRASTRUCTURE ONLY.

Functional (parameters are passed in, nothing is constructed) restatements of
  pytorch/ops/pt_custom_ops/pt_utils.py           MaskedQueryAndGroup :114-144, MaskedMaxPool :179-202,
                                                   MaskedUpsample :205-227
  pytorch/models/local_aggregation_operators.py   PosPool :16-112, AdaptiveWeight :115-224,
                                                   PointWiseMLP :227-316, PseudoGrid :319-426
on top of `oracle.native` (the C restatement of the five kernels).  Pinned by
tests/golden/operators_*.npz, which tests/golden/make_operator_golden.py produced by importing the
reference's own Python modules in the build container (tests/test_oracle_golden.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import native


class ExtCPU:
    """The reference's `pt_custom_ops._ext` function surface on CPU tensors, backed by oracle.native."""

    @staticmethod
    def group_points(points, idx):
        return torch.from_numpy(native.group_points(points.detach().numpy(), idx.numpy()))

    @staticmethod
    def group_points_grad(grad_out, idx, n):
        return torch.from_numpy(native.group_points_grad(grad_out.detach().numpy(), idx.numpy(), n))

    @staticmethod
    def masked_ordered_ball_query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample):
        i, m = native.masked_ordered_ball_query(query_xyz.numpy(), support_xyz.numpy(), query_mask.numpy(),
                                                support_mask.numpy(), radius, nsample)
        return [torch.from_numpy(i), torch.from_numpy(m)]

    @staticmethod
    def masked_grid_subsampling(points, mask, nsamples, sampleDl):
        s, m = native.masked_grid_subsampling(points.numpy(), mask.numpy(), nsamples, sampleDl)
        return [torch.from_numpy(s), torch.from_numpy(m)]

    @staticmethod
    def masked_nearest_query(query_xyz, support_xyz, query_mask, support_mask):
        i, m = native.masked_nearest_query(query_xyz.numpy(), support_xyz.numpy(), query_mask.numpy(),
                                           support_mask.numpy())
        return [torch.from_numpy(i), torch.from_numpy(m)]


class _Group(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.idx, ctx.n = idx, features.shape[2]
        return ExtCPU.group_points(features.contiguous(), idx)

    @staticmethod
    def backward(ctx, g):
        return ExtCPU.group_points_grad(g.contiguous(), ctx.idx, ctx.n), None


def query_and_group(query_xyz, support_xyz, query_mask, support_mask, features, radius, nsample,
                    normalize_xyz):
    """-> (grouped_features [B,C,M,K] or None, rel [B,3,M,K], idx_mask [B,M,K], idx)"""
    idx, idx_mask = ExtCPU.masked_ordered_ball_query(query_xyz, support_xyz, query_mask, support_mask,
                                                     radius, nsample)
    rel = _Group.apply(support_xyz.transpose(1, 2).contiguous(), idx)
    rel = rel - query_xyz.transpose(1, 2).unsqueeze(-1)
    if normalize_xyz:
        rel = rel / radius
    grouped = _Group.apply(features, idx) if features is not None else None
    return grouped, rel, idx_mask, idx


def reduce_neighbours(agg, reduction, idx_mask, query_mask):
    K = agg.shape[-1]
    if reduction == 'max':
        return F.max_pool2d(agg, kernel_size=[1, K]).squeeze(-1)
    fm = (idx_mask + (1 - query_mask[:, :, None]))[:, None]
    agg = agg * fm
    out = agg.sum(-1)
    if reduction in ('avg', 'mean'):
        out = out / fm.sum(-1)
    elif reduction != 'sum':
        raise NotImplementedError(reduction)
    return out


def pospool(query_xyz, support_xyz, query_mask, support_mask, features, radius, nsample,
            position_embedding='xyz', reduction='avg'):
    """PosPool up to (not including) the output transform.  -> [B,C,M]"""
    B, C, _ = features.shape
    M = query_xyz.shape[1]
    g, rel, idx_mask, _ = query_and_group(query_xyz, support_xyz, query_mask, support_mask, features,
                                          radius, nsample, True)
    if position_embedding == 'xyz':
        agg = (rel.unsqueeze(1) * g.view(B, C // 3, 3, M, nsample)).view(B, C, M, nsample)
    elif position_embedding == 'sin_cos':
        fd = C // 6
        rng = torch.arange(fd, dtype=torch.float32)
        dim_mat = torch.pow(1.0 * 1000, (1.0 / fd) * rng)
        div = torch.div((100 * rel).unsqueeze(-1), dim_mat)
        emb = torch.cat([torch.sin(div), torch.cos(div)], -1).permute(0, 1, 4, 2, 3).contiguous()
        agg = g * emb.view(B, C, M, nsample)
    else:
        raise NotImplementedError(position_embedding)
    return reduce_neighbours(agg, reduction, idx_mask, query_mask)


def adaptive_weight(query_xyz, support_xyz, query_mask, support_mask, features, radius, nsample,
                    conv_weights, conv_biases, shared_channels=1, reduction='avg'):
    """AdaptiveWeight 'dp' up to the output transform. conv_weights[i]: [Co,Ci] (1x1 conv), biases [Co]."""
    B, C, _ = features.shape
    M = query_xyz.shape[1]
    g, rel, idx_mask, _ = query_and_group(query_xyz, support_xyz, query_mask, support_mask, features,
                                          radius, nsample, True)
    w = rel
    for i, (W, b) in enumerate(zip(conv_weights, conv_biases)):
        if i > 0:
            w = torch.relu(w)
        w = F.conv2d(w, W.view(W.shape[0], W.shape[1], 1, 1), b)
    S = shared_channels
    agg = (g.view(B, C // S, S, M, nsample) * w.unsqueeze(2)).view(B, C, M, nsample)
    return reduce_neighbours(agg, reduction, idx_mask, query_mask)


def pointwise_mlp(query_xyz, support_xyz, query_mask, support_mask, features, radius, nsample,
                  layers, reduction='max', training=True, eps=1e-5):
    """PointWiseMLP 'dp_fi_df'.  layers: list of dicts {weight [Co,Ci], gamma, beta, running_mean,
    running_var}; BN2d uses batch statistics when training (as nn.BatchNorm2d does)."""
    g, rel, idx_mask, _ = query_and_group(query_xyz, support_xyz, query_mask, support_mask, features,
                                          radius, nsample, True)
    center = g[..., :1].expand(-1, -1, -1, nsample)
    x = torch.cat([rel, center, g - center], 1)
    for L in layers:
        W = L['weight']
        x = F.conv2d(x, W.view(W.shape[0], W.shape[1], 1, 1))
        x = F.batch_norm(x, L.get('running_mean'), L.get('running_var'), L['gamma'], L['beta'],
                         training=training, momentum=0.0, eps=eps)
        x = torch.relu(x)
    return reduce_neighbours(x, reduction, idx_mask, query_mask)


def pseudo_grid(query_xyz, support_xyz, query_mask, support_mask, features, radius, nsample, K_points,
                kernel_weights, extent, influence='linear'):
    """PseudoGrid up to the output transform."""
    B, C, _ = features.shape
    M = query_xyz.shape[1]
    P = K_points.shape[0]
    g, rel, idx_mask, _ = query_and_group(query_xyz, support_xyz, query_mask, support_mask, features,
                                          radius, nsample, False)
    diff = rel.permute(0, 2, 3, 1).unsqueeze(3) - K_points  # B,M,K,P,3
    sq = (diff ** 2).sum(-1)
    if influence == 'constant':
        w = torch.ones_like(sq)
    elif influence == 'linear':
        w = torch.clamp(1 - torch.sqrt(sq) / extent, min=0.0)
    else:
        raise ValueError(influence)
    w = w.permute(0, 1, 3, 2)
    fm = idx_mask + (1 - query_mask[:, :, None])
    w = (w * fm[:, :, None, :]).reshape(-1, P, nsample)
    nf = g.permute(0, 2, 3, 1).contiguous().view(-1, nsample, C)
    out = (torch.bmm(w, nf) * kernel_weights).sum(1)
    return out.view(B, M, C).transpose(1, 2)


def masked_max_pool(xyz, mask, features, npoint, radius, nsample, sampleDl):
    sub_xyz, sub_mask = ExtCPU.masked_grid_subsampling(xyz, mask, npoint, sampleDl)
    g, _, _, _ = query_and_group(sub_xyz, xyz, sub_mask, mask, features, radius, nsample, False)
    return sub_xyz, sub_mask, F.max_pool2d(g, kernel_size=[1, nsample]).squeeze(-1)


def masked_upsample_nearest(up_xyz, xyz, up_mask, mask, features):
    idx, _ = ExtCPU.masked_nearest_query(up_xyz, xyz, up_mask, mask)
    return _Group.apply(features, idx)[..., 0].contiguous()


def _conv_bn(x, state, prefix, relu, eps=1e-5):
    """Conv1d(1x1, no bias) + BatchNorm1d in training mode [+ ReLU] from a state dict (backbones/resnet.py:30-32)."""
    x = F.conv1d(x, state[prefix + "0.weight"])
    x = F.batch_norm(x, None, None, state[prefix + "1.weight"], state[prefix + "1.bias"], training=True, eps=eps)
    return torch.relu(x) if relu else x


def bottleneck(xyz, mask, features, state, kind, operator_kwargs, radius, nsample, in_channels, out_channels,
               downsample=False, sampleDl=None, npoint=None, eps=1e-5):
    """A whole residual `Bottleneck` in training mode (backbones/resnet.py:22-68) around one of the three gather-and-
    reduce operators: [MaskedMaxPool ->] conv1+BN+ReLU -> operator -> BN+ReLU (the operator's out_transform,
    local_aggregation_operators.py:43-45) -> conv2+BN (+ shortcut conv+BN) -> add -> ReLU.  `state`: the module's
    state dict as tensors (requires_grad where a gradient is wanted).  -> (query_xyz, query_mask, out)."""
    if downsample:
        q_xyz, q_mask, identity = masked_max_pool(xyz, mask, features, npoint, radius, nsample, sampleDl)
    else:
        q_xyz, q_mask, identity = xyz, mask, features
    x = _conv_bn(features, state, "conv1.", True, eps)
    a = (q_xyz, xyz, q_mask, mask, x, radius, nsample)
    pre = "local_aggregation.local_aggregation_operator."
    if kind == "pospool":
        y = pospool(*a, **operator_kwargs)
    elif kind == "adaptive_weight":
        y = adaptive_weight(*a, [state[pre + "mlps.conv0.weight"].flatten(1)], [state[pre + "mlps.conv0.bias"]],
                            **operator_kwargs)
    elif kind == "pseudo_grid":
        y = pseudo_grid(*a, state[pre + "K_points"], state[pre + "kernel_weights"], **operator_kwargs)
    else:
        raise NotImplementedError(kind)
    y = torch.relu(F.batch_norm(y, None, None, state[pre + "out_transform.0.weight"], state[pre + "out_transform.0.bias"],
                                training=True, eps=eps))
    y = _conv_bn(y, state, "conv2.", False, eps)
    if in_channels != out_channels:
        identity = _conv_bn(identity, state, "shortcut.", False, eps)
    return q_xyz, q_mask, torch.relu(y + identity)


# ------------------------------------------------------------------ synthetic clouds
def make_cloud(rng, B, N, kind='uniform', pad_frac=0.0):
    """xyz [B,N,3] f32, mask [B,N] i32 with the dataset contract: valid points first, padding last,
    padding coordinates copied from valid points (datasets/ModelNet40.py:186-196)."""
    if kind == 'uniform':
        xyz = rng.random((B, N, 3), dtype=np.float32)
    elif kind == 'planes':
        xyz = np.empty((B, N, 3), np.float32)
        for b in range(B):
            pl = rng.integers(0, 6, N)
            u = rng.random((N, 3), dtype=np.float32)
            for i in range(6):
                sel = pl == i
                axis = i % 3
                u[sel, axis] = (0.15 + 0.14 * i) + 0.01 * rng.standard_normal(sel.sum()).astype(np.float32)
            xyz[b] = u
    else:
        raise ValueError(kind)
    mask = np.ones((B, N), np.int32)
    if pad_frac > 0:
        nvalid = max(1, int(round(N * (1 - pad_frac))))
        for b in range(B):
            for i in range(nvalid, N):
                xyz[b, i] = xyz[b, i % nvalid]
            mask[b, nvalid:] = 0
    return xyz, mask

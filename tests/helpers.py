"""Shared test helpers: fixture loading, config objects, oracle-side evaluation of an operator fixture."""
import ast
import glob
import os

import numpy as np
import torch
import torch.nn as nn


def _oo():
    from oracle import operators
    return operators

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


class AttrDict(dict):
    """Minimal stand-in for the reference's EasyDict config object."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def default_config(kind, over=None, **extra):
    """Field names and defaults of the reference's utils/config.py:27-103 that the operators read."""
    cfg = AttrDict(
        bn_momentum=0.1, density_parameter=5.0, local_aggregation_type=kind,
        pospool=AttrDict(position_embedding='xyz', reduction='sum', output_conv=False),
        adaptive_weight=AttrDict(weight_type='dp', num_mlps=1, shared_channels=1, weight_softmax=False,
                                 reduction='avg', output_conv=False),
        pointwisemlp=AttrDict(feature_type='dp_fj', num_mlps=1, reduction='max'),
        pseudo_grid=AttrDict(fixed_kernel_points='center', KP_influence='linear', KP_extent=1.0,
                             num_kernel_points=15, convolution_mode='sum', output_conv=False))
    for k, v in (over or {}).items():
        sub, _, leaf = k.partition("__")
        if leaf:
            cfg[sub][leaf] = v
        else:
            cfg[k] = v
    cfg.update(extra)
    return cfg


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    fx = {k: z[k] for k in z.files}
    if "over" in fx:
        fx["over"] = ast.literal_eval(str(fx["over"]))
    if "kind" in fx:
        fx["kind"] = str(fx["kind"])
    return fx


def operator_fixtures():
    names = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "operators_*.npz")))
    return [n for n in names if "resnet" not in n and "bottleneck" not in n]


def state_of(fx, prefix=""):
    out = {}
    for k, v in fx.items():
        if k.startswith("state__" + prefix):
            out[k[len("state__" + prefix):]] = torch.from_numpy(np.array(v))
    return out


def assert_close(a, b, tol=1e-5, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b)
    bound = tol + tol * np.abs(b)
    if not (err <= bound).all():
        i = np.unravel_index(np.argmax(err - bound), err.shape)
        raise AssertionError(f"{what}: max violation at {i}: got {a[i]!r} want {b[i]!r} (|err| {err[i]:.3e}, tol {tol})")


# ---- oracle-side evaluation of an operator "fixture" (a dict with inputs, state__*, kind, over, ...)
def _output_transform(state, x, training):
    """out_conv / out_transform of the operator, rebuilt from the fixture's state dict."""
    pre = "local_aggregation_operator."
    if pre + "out_conv.0.weight" in state:
        w = state[pre + "out_conv.0.weight"]
        x = torch.nn.functional.conv1d(x, w)
        bn = pre + "out_conv.1."
    elif pre + "out_transform.0.weight" in state:
        bn = pre + "out_transform.0."
    else:
        return x
    x = torch.nn.functional.batch_norm(x, state[bn + "running_mean"].clone(), state[bn + "running_var"].clone(),
                                       state[bn + "weight"], state[bn + "bias"], training=training,
                                       momentum=0.1, eps=1e-5)
    return torch.relu(x)


def oracle_operator(fx):
    """Evaluate one operator fixture with the oracle; returns (out, grad_features, param grads)."""
    st = {k: v.clone().requires_grad_(v.dtype == torch.float32 and "running" not in k) for k, v in state_of(fx).items()}
    xyz = torch.from_numpy(fx["xyz"])
    mask = torch.from_numpy(fx["mask"])
    feats = torch.from_numpy(fx["features"]).clone().requires_grad_(True)
    radius, K = float(fx["radius"]), int(fx["nsample"])
    training = bool(fx["training"])
    kind, over = fx["kind"], fx["over"]
    pre = "local_aggregation_operator."
    a = (xyz, xyz, mask, mask, feats, radius, K)
    if kind == "pospool":
        y = _oo().pospool(*a, position_embedding=over["pospool__position_embedding"], reduction=over["pospool__reduction"])
    elif kind == "adaptive_weight":
        n = over.get("adaptive_weight__num_mlps", 1)
        Ws = [st[pre + f"mlps.conv{i}.weight"].flatten(1) for i in range(n)]
        bs = [st[pre + f"mlps.conv{i}.bias"] for i in range(n)]
        y = _oo().adaptive_weight(*a, Ws, bs, shared_channels=over.get("adaptive_weight__shared_channels", 1),
                               reduction=over["adaptive_weight__reduction"])
    elif kind == "pointwisemlp":
        n = over["pointwisemlp__num_mlps"]
        layers = []
        for i in range(n):
            p = pre + f"mlps.conv{i}."
            layers.append(dict(weight=st[p + "0.weight"].flatten(1), gamma=st[p + "1.weight"], beta=st[p + "1.bias"],
                               running_mean=st[p + "1.running_mean"].clone(), running_var=st[p + "1.running_var"].clone()))
        y = _oo().pointwise_mlp(*a, layers, reduction=over["pointwisemlp__reduction"], training=training)
    elif kind == "pseudo_grid":
        extent = 2 * 1.0 * radius / 5.0
        y = _oo().pseudo_grid(*a, st[pre + "K_points"], st[pre + "kernel_weights"], extent,
                           influence=over.get("pseudo_grid__KP_influence", "linear"))
    else:
        raise AssertionError(kind)
    if kind != "pointwisemlp":
        y = _output_transform(st, y, training)
    (y * torch.from_numpy(fx["probe"])).sum().backward()
    grads = {k: v.grad for k, v in st.items() if v.requires_grad and v.grad is not None}
    return y.detach(), feats.grad, grads



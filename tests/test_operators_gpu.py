"""GPU parity of the operator API: closerlook3d_amd's LocalAggregation / Bottleneck / ResNet / seg head
loaded with the fixture's state dict (same parameter names as the reference) against the golden
vectors produced by the reference's own Python modules.  Tolerance: 1e-5 (north_star) on outputs,
slightly looser on gradients that accumulate over B*M*K terms."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close, default_config, load_fixture, operator_fixtures, state_of

pytestmark = pytest.mark.gpu

IMPLS = ["grouped", "auto"]


def _build_la(fx, impl):
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    C = fx["features"].shape[1]
    cfg = default_config(fx["kind"], fx["over"], cl3d_impl=impl)
    mod = LocalAggregation(C, C, float(fx["radius"]), int(fx["nsample"]), cfg)
    missing = mod.load_state_dict(state_of(fx), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return mod.cuda().train(bool(fx["training"]))


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("name", operator_fixtures())
def test_local_aggregation_matches_reference(name, impl):
    fx = load_fixture(name)
    mod = _build_la(fx, impl)
    xyz = torch.from_numpy(fx["xyz"]).cuda()
    mask = torch.from_numpy(fx["mask"]).cuda()
    feats = torch.from_numpy(fx["features"]).cuda().requires_grad_(True)
    out = mod(xyz, xyz, mask, mask, feats)
    (out * torch.from_numpy(fx["probe"]).cuda()).sum().backward()
    assert_close(out.detach().cpu().numpy(), fx["out"], 1e-5, f"{name}[{impl}] out")
    assert_close(feats.grad.cpu().numpy(), fx["grad_features"], 3e-5, f"{name}[{impl}] grad_features")
    for k, p in mod.named_parameters():
        if "grad__" + k in fx:
            assert p.grad is not None, k
            assert_close(p.grad.cpu().numpy(), fx["grad__" + k], 1e-4, f"{name}[{impl}] grad {k}")


@pytest.mark.parametrize("impl", IMPLS)
def test_strided_bottleneck_matches_reference(impl):
    from closerlook3d_amd.backbones import Bottleneck
    fx = load_fixture("operators_strided_bottleneck.npz")
    cfg = default_config("pospool", {"pospool__position_embedding": "xyz", "pospool__reduction": "avg"}, cl3d_impl=impl)
    btn = Bottleneck(24, 48, 2, 0.15, 16, cfg, downsample=True, sampleDl=0.12, npoint=64)
    btn.load_state_dict(state_of(fx), strict=True)
    btn = btn.cuda().train(True)
    feats = torch.from_numpy(fx["features"]).cuda().requires_grad_(True)
    sub_xyz, sub_mask, out = btn(torch.from_numpy(fx["xyz"]).cuda(), torch.from_numpy(fx["mask"]).cuda(), feats)
    assert np.array_equal(sub_xyz.cpu().numpy().view(np.uint32), fx["out0"].view(np.uint32))
    assert np.array_equal(sub_mask.cpu().numpy(), fx["out1"])
    (out * torch.from_numpy(fx["probe"]).cuda()).sum().backward()
    assert_close(out.detach().cpu().numpy(), fx["out"], 1e-5, "bottleneck out")
    assert_close(feats.grad.cpu().numpy(), fx["grad_features"], 5e-5, "bottleneck grad_features")


@pytest.mark.parametrize("kind", ["pospool", "pointwisemlp"])
def test_resnet_and_seg_head_match_reference(kind):
    """5-stage backbone + nearest-upsampling decode on a seeded cloud (the integration pin, 8(c)(iv))."""
    from closerlook3d_amd.backbones import ResNet, SceneSegHeadResNet
    from closerlook3d_amd.pt_utils import ball_query_cache
    fx = load_fixture(f"operators_resnet_seg_{kind}.npz")
    cfg = default_config(kind, fx["over"])
    K = 16
    net = ResNet(cfg, 3, 0.1, 0.05, [K] * 5, [128, 48, 16, 8], width=12, depth=2, bottleneck_ratio=2)
    head = SceneSegHeadResNet(5, 12, 0.1, [K] * 5)
    net.load_state_dict(state_of(fx, "backbone."), strict=True)
    head.load_state_dict(state_of(fx, "head."), strict=True)
    net, head = net.cuda().train(True), head.cuda().train(True)
    feats = torch.from_numpy(fx["features"]).cuda().requires_grad_(True)
    with ball_query_cache():
        ep = net(torch.from_numpy(fx["xyz"]).cuda(), torch.from_numpy(fx["mask"]).cuda(), feats)
        logits = head(ep)
    assert np.array_equal(ep["res5_xyz"].cpu().numpy().view(np.uint32), fx["out0"].view(np.uint32))
    assert np.array_equal(ep["res5_mask"].cpu().numpy(), fx["out1"])
    assert np.array_equal(ep["res3_xyz"].cpu().numpy().view(np.uint32), fx["out3"].view(np.uint32))
    # ten BN layers deep: allow float noise to grow a little beyond the single-operator bound
    assert_close(ep["res5_features"].detach().cpu().numpy(), fx["out2"], 2e-4, "res5_features")
    assert_close(logits.detach().cpu().numpy(), fx["out"], 2e-4, "logits")
    (logits * torch.from_numpy(fx["probe"]).cuda()).sum().backward()
    assert_close(feats.grad.cpu().numpy(), fx["grad_features"], 1e-3, "d logits / d input features")

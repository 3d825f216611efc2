"""GPU parity of the operator API: closerlook3d_amd's LocalAggregation / Bottleneck / ResNet / seg head
loaded with the fixture's state dict (same parameter names as the reference) against the golden
vectors produced by the reference's own Python modules.  Tolerance: 1e-5 (north_star) on outputs,
slightly looser on gradients that accumulate over B*M*K terms."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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
            # sums of B*M*K products taken in a different order than the reference; on a cold MIOpen cache the library
            # may also pick another convolution algorithm for the 'grouped' dataflow (observed once: 1.2e-4)
            assert_close(p.grad.cpu().numpy(), fx["grad__" + k], 2e-4, f"{name}[{impl}] grad {k}")


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
    # The input gradient goes through ten layers of arg-max routing (max over neighbours, max-pool): float32 noise
    # re-routes near-ties and moves whole entries -- in the reference's own float32 run as well.  It is therefore held
    # against the float64 anchor, entry count against the reference's own count, in
    # tests/test_fp64_anchor_gpu.py::test_resnet_input_gradient_is_as_close_to_the_anchor_as_the_reference.
    (logits * torch.from_numpy(fx["probe"]).cuda()).sum().backward()
    assert torch.isfinite(feats.grad).all()


@pytest.mark.parametrize("kind", ["pospool", "pointwisemlp"])
def test_geometry_prefetch_on_index_streams_changes_nothing(kind, monkeypatch):
    """The backbone with the coordinates-only work (subsampling pyramid, ball queries, CSR builds) forked onto the
    index streams -- the mode a captured step runs in -- against the same step on one stream: identical bits, and
    every prefetched product is the one the layers ask for (no second search under another key)."""
    from closerlook3d_amd import pt_utils
    from closerlook3d_amd.backbones import ResNet
    fx = load_fixture(f"operators_resnet_seg_{kind}.npz")
    cfg = default_config(kind, fx["over"])
    K = 16
    net = ResNet(cfg, 3, 0.1, 0.05, [K] * 5, [128, 48, 16, 8], width=12, depth=2, bottleneck_ratio=2)
    net.load_state_dict(state_of(fx, "backbone."), strict=True)
    net = net.cuda().train(True)
    xyz, mask = torch.from_numpy(fx["xyz"]).cuda(), torch.from_numpy(fx["mask"]).cuda()
    probe = None
    res = {}
    for mode in (False, True):
        monkeypatch.setattr(pt_utils, "ASYNC_INDEX", mode)
        monkeypatch.setattr(pt_utils, "PREFETCH_GEOMETRY", True)
        net.zero_grad(set_to_none=True)
        feats = torch.from_numpy(fx["features"]).cuda().requires_grad_(True)
        with pt_utils.ball_query_cache():
            ep = net(xyz, mask, feats)
            entries = len(pt_utils._BQ_CACHE)
        out = ep["res5_features"]
        if probe is None:
            probe = torch.randn_like(out)
        out.backward(probe)
        torch.cuda.synchronize()
        res[mode] = (entries, out.detach().clone(), ep["res5_xyz"].clone(), feats.grad.clone(),
                     [p.grad.clone() for p in net.parameters() if p.grad is not None])
    assert res[True][0] == res[False][0] == 9 + 4, "prefetch keys must be the keys the layers look up"
    for a, b in zip(res[True][1:3], res[False][1:3]):
        assert torch.equal(a, b)
    # the callers' library convolutions may sum their weight gradients in a run-dependent order
    for a, b in zip([res[True][3]] + res[True][4], [res[False][3]] + res[False][4]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12


# ---------------------------------------------------------------- fused kernels at realistic widths
FUSED_CASES = [
    # kind, overrides, C, K, N, in-radius multiple
    ("pospool", {"pospool__position_embedding": "xyz", "pospool__reduction": "avg"}, 72, 32, 1024, 1.5),
    ("pospool", {"pospool__position_embedding": "sin_cos", "pospool__reduction": "avg"}, 72, 16, 512, 4.0),
    ("pospool", {"pospool__position_embedding": "xyz", "pospool__reduction": "sum"}, 9, 20, 300, 1.5),   # V=1 path
    ("adaptive_weight", {"adaptive_weight__num_mlps": 1, "adaptive_weight__reduction": "avg"}, 64, 32, 1024, 1.5),
    ("adaptive_weight", {"adaptive_weight__num_mlps": 1, "adaptive_weight__shared_channels": 4,
                         "adaptive_weight__reduction": "sum"}, 144, 23, 512, 4.0),
    ("pointwisemlp", {"pointwisemlp__feature_type": "dp_fi_df", "pointwisemlp__num_mlps": 1,
                      "pointwisemlp__reduction": "max"}, 64, 32, 1024, 1.5),
    ("pointwisemlp", {"pointwisemlp__feature_type": "dp_fi_df", "pointwisemlp__num_mlps": 1,
                      "pointwisemlp__reduction": "max"}, 18, 9, 300, 4.0),                              # V=1 path
    ("pointwisemlp", {"pointwisemlp__feature_type": "dp_fi_df", "pointwisemlp__num_mlps": 1,
                      "pointwisemlp__reduction": "max"}, 288, 16, 256, 4.0),   # deep-stage width: channel chunks over gridDim.y
    ("pseudo_grid", {"pseudo_grid__KP_influence": "linear"}, 64, 26, 1024, 1.5),
    ("pseudo_grid", {"pseudo_grid__KP_influence": "constant"}, 36, 16, 400, 4.0),
    ("pseudo_grid", {"pseudo_grid__KP_influence": "linear"}, 144, 20, 256, 4.0),    # channel chunks over gridDim.y
    ("pseudo_grid", {"pseudo_grid__KP_influence": "linear"}, 12, 26, 20000, 1.5),   # scene-sized support set
]


@pytest.mark.parametrize("kind,over,C,K,N,mult", FUSED_CASES)
def test_fused_operator_matches_oracle(kind, over, C, K, N, mult):
    """impl='fused' (must not fall back) against the CPU oracle, forward and all gradients."""
    _fused_vs_oracle(kind, over, C, K, N, mult)


def test_pseudo_grid_with_many_influences_per_slot():
    """The sparse PseudoGrid form keeps four (kernel point, influence) pairs per slot and sums a slot with more of them
    densely in place.  At the reference's kernel-point spacing no slot has more than four; kernel points pulled in to
    0.3 of their radius overlap heavily, so most slots take that path (forward, support-major backward and the
    d kernel_weights staging)."""
    def pull_in(mod):
        with torch.no_grad():
            mod.local_aggregation_operator.K_points.mul_(0.3)
    _fused_vs_oracle("pseudo_grid", {"pseudo_grid__KP_influence": "linear"}, 32, 20, 600, 1.5, tweak=pull_in)


def _fused_vs_oracle(kind, over, C, K, N, mult, tweak=None):
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    from oracle import operators as oo
    from tests.helpers import oracle_operator
    B = 2
    rng = np.random.default_rng(C * 1000 + K)
    xyz, mask = oo.make_cloud(rng, B, N, kind="uniform", pad_frac=0.0)
    x1, m1 = oo.make_cloud(rng, 1, N, kind="planes", pad_frac=0.2)
    xyz[-1], mask[-1] = x1[0], m1[0]
    feats = rng.standard_normal((B, C, N)).astype(np.float32)
    radius = float((mult * K * 3 / (4 * np.pi * N)) ** (1 / 3))
    torch.manual_seed(C + K)
    mod = LocalAggregation(C, C, radius, K, default_config(kind, over, cl3d_impl="fused"))
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
    if tweak is not None:
        tweak(mod)
    state = {"state__" + k: v.detach().clone().numpy() for k, v in mod.state_dict().items()}
    probe = rng.standard_normal((B, C, N)).astype(np.float32)
    fx = dict(state, xyz=xyz, mask=mask, features=feats, radius=np.float32(radius), nsample=np.int32(K),
              training=np.int32(1), kind=kind, over=over, probe=probe)
    want, want_gf, want_grads = oracle_operator(fx)

    mod = mod.cuda().train(True)
    t_xyz, t_mask = torch.from_numpy(xyz).cuda(), torch.from_numpy(mask).cuda()
    f = torch.from_numpy(feats).cuda().requires_grad_(True)
    out = mod(t_xyz, t_xyz, t_mask, t_mask, f)
    (out * torch.from_numpy(probe).cuda()).sum().backward()
    assert_close(out.detach().cpu().numpy(), want.numpy(), 1e-5, f"{kind} fused out")
    assert_close(f.grad.cpu().numpy(), want_gf.numpy(), 5e-5, f"{kind} fused grad_features")
    for k, p in mod.named_parameters():
        if k in want_grads:
            # parameter gradients are fp32 sums of B*M*K products with heavy cancellation, taken in a different
            # order by the engine (per-lane partials, per-block partials, fixed-order tree) and the oracle
            scale = float(want_grads[k].abs().max()) + 1e-12
            assert_close(p.grad.cpu().numpy() / scale, want_grads[k].numpy() / scale, 1e-4, f"{kind} fused grad {k}")
    # run-to-run repeatability of the fused backward (ordered gathers, no float atomics)
    f2 = torch.from_numpy(feats).cuda().requires_grad_(True)
    mod.zero_grad()
    out2 = mod(t_xyz, t_xyz, t_mask, t_mask, f2)
    (out2 * torch.from_numpy(probe).cuda()).sum().backward()
    assert torch.equal(out, out2) and torch.equal(f.grad, f2.grad)


@pytest.mark.parametrize("B,N,M,K", [(3, 500, 321, 17), (2, 4096, 4096, 32), (1, 20000, 5000, 16), (1, 40000, 3000, 8),
                                     (1, 40960, 10240, 31), (2, 81920, 20480, 26), (1, 32768, 1000, 5), (1, 32769, 1000, 5),
                                     (3, 100000, 777, 4)])
def test_inverse_index_matches_numpy(B, N, M, K):
    """CSR inverse (wave-private counting sort; N > 32 768: the same kernels over key ranges of <= 16 384 support indices --
    configs 3 / 5's scenes, round 6: no library sort) == numpy's stable argsort; indices outside [0,N) are dropped; the
    table is identical build after build."""
    from closerlook3d_amd.fused import inverse_index
    rng = np.random.default_rng(4)
    idx = rng.integers(0, N, (B, M, K)).astype(np.int32)
    idx[0, :, :] = 7  # one support point referenced by every slot of cloud 0 (one very long segment)
    if B > 1:
        idx[1, ::5, 3] = -1  # invalid indices are not part of any row
        idx[1, 1::7, 0] = N
    tables = []
    for _ in range(2):
        off, slots = inverse_index(torch.from_numpy(idx).cuda(), N)
        tables.append((off.cpu().numpy(), slots.cpu().numpy()))
    assert np.array_equal(tables[0][0], tables[1][0]) and all(
        np.array_equal(tables[0][1][b, :tables[0][0][b, N]], tables[1][1][b, :tables[1][0][b, N]]) for b in range(B))
    off, slots = tables[0]
    for b in range(B):
        flat = idx[b].reshape(-1)
        valid = np.nonzero((flat >= 0) & (flat < N))[0]
        order = valid[np.argsort(flat[valid], kind="stable")]
        counts = np.bincount(flat[valid], minlength=N)
        assert np.array_equal(off[b], np.concatenate([[0], np.cumsum(counts)]).astype(np.int32))
        assert np.array_equal(slots[b, :off[b, N]], order.astype(np.int32))


def test_eval_mode_inference_paths_agree():
    """eval-mode (running statistics) forward: fused == grouped for every operator kind."""
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    from oracle import operators as oo
    rng = np.random.default_rng(8)
    B, N, K, C = 2, 512, 16, 24
    xyz, mask = oo.make_cloud(rng, B, N, pad_frac=0.1)
    feats = rng.standard_normal((B, C, N)).astype(np.float32)
    t = [torch.from_numpy(a).cuda() for a in (xyz, xyz, mask, mask, feats)]
    for kind, over in (("pospool", {"pospool__reduction": "avg"}), ("adaptive_weight", {}),
                       ("pointwisemlp", {"pointwisemlp__feature_type": "dp_fi_df"}), ("pseudo_grid", {})):
        torch.manual_seed(1)
        a = LocalAggregation(C, C, 0.2, K, default_config(kind, over, cl3d_impl="fused")).cuda().eval()
        b = LocalAggregation(C, C, 0.2, K, default_config(kind, over, cl3d_impl="grouped")).cuda().eval()
        b.load_state_dict(a.state_dict())
        with torch.no_grad():
            assert_close(a(*t).cpu().numpy(), b(*t).cpu().numpy(), 1e-5, f"{kind} eval fused vs grouped")


@pytest.mark.parametrize("targets", [True, False])
@pytest.mark.parametrize("C,K,N,npoint", [(24, 16, 512, 128), (144, 32, 2048, 512), (10, 9, 300, 77), (8, 24, 9000, 2252)])
def test_fused_max_pool_matches_reference_dataflow(C, K, N, npoint, targets, monkeypatch):
    """MaskedMaxPool: fused kernel == gather + F.max_pool2d (values and gradient routing, incl. ReLU-zero ties), in both
    forms of the engine's pooling: the arg-max kept as a support index with the gradient scattered (round 6, the default;
    9000 points: three support tiles of the scatter's LDS rows, 77 / 2252 queries: not a multiple of four) and the
    arg-max kept as a slot byte with the ordered gather through the CSR inverse."""
    from closerlook3d_amd import fused as _fused
    from closerlook3d_amd.pt_utils import MaskedMaxPool
    from oracle import operators as oo
    monkeypatch.setattr(_fused, "MAXPOOL_TARGETS", targets)
    rng = np.random.default_rng(C + K)
    B = 2
    xyz, mask = oo.make_cloud(rng, B, N, pad_frac=0.15)
    feats = np.maximum(rng.standard_normal((B, C, N)), 0).astype(np.float32)  # many exact ties at 0
    t_xyz, t_mask = torch.from_numpy(xyz).cuda(), torch.from_numpy(mask).cuda()
    outs = []
    for fused in (True, False):
        pool = MaskedMaxPool(npoint, 0.15, K, 0.08).cuda()
        pool.fused = fused
        f = torch.from_numpy(feats).cuda().requires_grad_(True)
        sub_xyz, sub_mask, y = pool(t_xyz, t_mask, f)
        probe = torch.from_numpy(rng.standard_normal((B, C, npoint)).astype(np.float32)).cuda() if not outs else outs[0][3]
        (y * probe).sum().backward()
        outs.append((y.detach(), f.grad.clone(), sub_xyz, probe))
    assert torch.equal(outs[0][2], outs[1][2])
    assert torch.equal(outs[0][0], outs[1][0])
    assert_close(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy(), 1e-5, "max-pool input gradient")
    # and against the CPU oracle
    want = oo.masked_max_pool(torch.from_numpy(xyz), torch.from_numpy(mask), torch.from_numpy(feats), npoint, 0.15, K, 0.08)
    assert np.array_equal(outs[0][0].cpu().numpy(), want[2].numpy())


@pytest.mark.parametrize("B,C,N,Co", [(4, 64, 512, 64), (3, 10, 77, 7), (8, 300, 40, 300), (16, 576, 32, 576),
                                      (2, 1152, 8, 1152)])   # the last one takes the 64 x 16 merge tiles
@pytest.mark.parametrize("engine", ["mfma", "library"])
def test_point_rows_weight_plumbing_matches_autograd(B, C, N, Co, engine):
    """The per-point GEMM of PointWiseMLP -- the engine's MFMA kernel with its weight split / ordered partial reduce,
    and the vendor-library variant kept for the A/B script with its split / merge kernels (both the element-per-thread
    merge and the tiled one used once the per-cloud products outgrow the L2s) -- against the same algebra in plain
    autograd."""
    from closerlook3d_amd.fused import _PointRows
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from ab.library_arms import PointRowsLibrary as _PointRowsLibrary  # (the library arm lives outside the package)
    torch.manual_seed(C + Co)
    f = torch.randn(B, C, N, device="cuda")
    W = torch.randn(Co, 3 + 2 * C, device="cuda") / np.sqrt(C)
    g_rows = torch.randn(B, N, 2 * Co, device="cuda")
    g_wr = torch.randn(Co, 3, device="cuda")
    res = []
    for mine in (True, False):
        fi, Wi = f.clone().requires_grad_(True), W.clone().requires_grad_(True)
        if mine:
            rows, wr = _PointRows.apply(fi, Wi, 0) if engine == "mfma" else _PointRowsLibrary.apply(fi, Wi)
        else:
            wr, wc, wd = Wi[:, :3], Wi[:, 3:3 + C], Wi[:, 3 + C:]
            rows = torch.einsum("bcn,oc->bno", fi, torch.cat([wd, wc - wd], 0))
        ((rows * g_rows).sum() + (wr * g_wr).sum()).backward()
        res.append((rows.detach(), wr.detach(), fi.grad, Wi.grad))
    for name, a, b in zip(("rows", "wr", "dfeat", "dW"), *res):
        scale = float(b.abs().max())
        assert_close(a.cpu().numpy() / scale, b.cpu().numpy() / scale, 2e-5, f"point rows {name}")


@pytest.mark.parametrize("B,C,N", [(16, 72, 4096), (2, 10, 301), (1, 3, 20000), (3, 144, 64)])
@pytest.mark.parametrize("training", [True, False])
def test_bn_relu_matches_torch_modules(B, C, N, training):
    """The engine's BatchNorm1d + ReLU against nn.BatchNorm1d + nn.ReLU: output, gradients (training), running
    statistics and num_batches_tracked."""
    from closerlook3d_amd import fused
    torch.manual_seed(B * 100 + C)
    ref = torch.nn.BatchNorm1d(C, momentum=0.1).cuda()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.weight[0] = -0.7  # a negative gamma
        ref.bias.normal_(0, 0.3)
        ref.running_mean.normal_(0, 0.2)
        ref.running_var.uniform_(0.5, 2.0)
    mine = torch.nn.BatchNorm1d(C, momentum=0.1).cuda()
    mine.load_state_dict(ref.state_dict())
    ref.train(training)
    mine.train(training)
    x = (torch.randn(B, C, N, device="cuda") * 1.7 + 0.4)
    g = torch.randn(B, C, N, device="cuda")
    xr = x.clone().requires_grad_(training)
    xm = x.clone().requires_grad_(training)
    with torch.set_grad_enabled(training):
        want = torch.relu(ref(xr))
        got = fused.bn_relu(xm, mine)
    assert got is not None
    assert_close(got.detach().cpu().numpy(), want.detach().cpu().numpy(), 1e-5, "bn_relu out")
    if training:
        want.backward(g)
        got.backward(g)
        scale = float(xr.grad.abs().max())
        assert_close(xm.grad.cpu().numpy() / scale, xr.grad.cpu().numpy() / scale, 2e-5, "bn_relu dx")
        for name in ("weight", "bias"):
            a, b = getattr(mine, name).grad, getattr(ref, name).grad
            assert ((a - b).norm() / b.norm()).item() < 1e-4, name
        assert_close(mine.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy(), 1e-5, "running_mean")
        assert_close(mine.running_var.cpu().numpy(), ref.running_var.cpu().numpy(), 1e-5, "running_var")
        assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == 1


@pytest.mark.parametrize("B,C,N", [(16, 72, 4096), (70, 8, 40000), (5, 144, 1000)])
def test_bn_statistics_finished_inside_the_launch_repeat_bit_for_bit(B, C, N):
    """Round 6: the workgroup that arrives last at a channel's ticket finishes the channel's statistics (no finalize
    launch).  Whichever workgroup that is, the partial blocks are added in block order: 100 eager passes and 100
    replays of a captured pass over the same input give the same bits, forward (scale/shift seen through the output,
    running statistics) and backward (dx, d gamma, d beta).  (70, 8, 40000): 210 partial blocks per channel, more
    than the 64 lanes that add them.)"""
    from closerlook3d_amd import fused
    import closerlook3d_amd
    torch.manual_seed(B + C)
    bn = torch.nn.BatchNorm1d(C, momentum=0.1).cuda().train(True)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
    x = (torch.randn(B, C, N, device="cuda") * 1.7 + 0.4).requires_grad_(True)
    g = torch.randn(B, C, N, device="cuda")

    def one_pass():
        with torch.no_grad():
            bn.running_mean.zero_()
            bn.running_var.fill_(1.0)
        x.grad = bn.weight.grad = bn.bias.grad = None
        out = fused.bn_relu(x, bn)
        out.backward(g)
        return [t.detach().clone() for t in (out, x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean, bn.running_var)]

    st = closerlook3d_amd.step_stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        first = one_pass()
        for _ in range(100):
            for a, b in zip(one_pass(), first):
                assert torch.equal(a, b)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            held = one_pass()
        for _ in range(100):
            with torch.no_grad():
                bn.running_mean.zero_()
                bn.running_var.fill_(1.0)
            graph.replay()
            for a, b in zip(held[:4] + [bn.running_mean, bn.running_var], first):
                assert torch.equal(a, b)
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()


@pytest.mark.parametrize("rel,B,N", [("modelnet/pointwisemlp_dp_fi_df_fc1.yaml", 2, 2048),
                                     ("partnet/adaptiveweight_dp_fc1_avg.yaml", 2, 2500),
                                     ("s3dis/pseudo_grid.yaml", 1, 8000)])
def test_reference_configs_build_and_train_one_step(rel, B, N):
    """compat.build_model on the reference's own option trees (golden JSON of its YAML files): one forward +
    backward + SGD step on the GPU for each task's model wrapper, at the configuration's own K / stage sizes."""
    import json
    import os
    from closerlook3d_amd import compat
    from closerlook3d_amd.pt_utils import ball_query_cache
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_configs.json")) as fh:
        cfg = compat.Config(json.load(fh)["merged"][rel])
    if rel.startswith("partnet"):
        cfg.num_parts, cfg.num_classes = [4, 2, 6], 3
    cfg.width = 36  # narrower than the shipped 144 to keep the test quick; everything else as configured
    cfg.npoints = [max(16, n * N // cfg.num_points) for n in cfg.npoints]
    torch.manual_seed(0)
    model = compat.build_model(cfg).cuda().train(True)
    model.init_weights()
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    rng = np.random.default_rng(3)
    scale = 1.0 if rel.startswith("modelnet") or rel.startswith("partnet") else 3.0
    xyz = torch.from_numpy((rng.random((B, N, 3)) * scale).astype(np.float32)).cuda()
    mask = torch.ones(B, N, dtype=torch.int32, device="cuda")
    feats = torch.randn(B, cfg.input_features_dim, N, device="cuda")
    with ball_query_cache():
        out = model(xyz, mask, feats)
    outs = out if isinstance(out, list) else [out]
    want = {"resnet_cls": [(B, cfg.num_classes)], "resnet_scene_seg": [(B, cfg.num_classes, N)],
            "resnet_part_seg": [(B, p, N) for p in (cfg.num_parts if isinstance(cfg.num_parts, list) else [])]}[cfg.head]
    assert [tuple(o.shape) for o in outs] == want
    sum(o.square().mean() for o in outs).backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    opt.step()


# ---------------------------------------------------------------- bf16 contraction (BASELINE config 2)
# Declared tolerance of the bf16 variant: the inputs of the dense contraction (features and the [W_d ; W_c - W_d]
# weight, K = C terms per output) are rounded to bf16 (8-bit mantissa, relative step 2^-8) on their way to the matrix
# cores; products and sums are f32; coordinates, indices, W_r * rel, BatchNorm statistics and the max stay f32.
BF16_REL_L2 = 1e-2     # relative L2 error of an operator's output against the f32 reference values
BF16_MAX = 4e-2        # largest element error relative to the largest magnitude
BF16_GRAD_REL_L2 = 8e-2  # gradients: the max over K routes each (query, channel) gradient to ONE neighbour; rounding
#                          noise of 2^-9 flips that choice for near-ties, which moves whole gradient terms between rows
#                          (measured at the metric shape: 4.6e-2 on d features)


def _rel_l2(got, want):
    got, want = got.astype(np.float64), want.astype(np.float64)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


def _rel_max(got, want):
    return float(np.abs(got.astype(np.float64) - want).max() / max(np.abs(want).max(), 1e-30))


def test_pointwisemlp_bf16_contraction_against_f32_reference():
    """config 2's operator with the contraction in bf16 against the reference-generated f32 fixture."""
    name = "operators_pointwisemlp_fc1_max_train.npz"
    fx = load_fixture(name)
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    C = fx["features"].shape[1]
    cfg = default_config(fx["kind"], fx["over"], cl3d_impl="fused", cl3d_precision="bf16")
    mod = LocalAggregation(C, C, float(fx["radius"]), int(fx["nsample"]), cfg)
    mod.load_state_dict(state_of(fx), strict=True)
    mod = mod.cuda().train(True)
    xyz, mask = torch.from_numpy(fx["xyz"]).cuda(), torch.from_numpy(fx["mask"]).cuda()
    feats = torch.from_numpy(fx["features"]).cuda().requires_grad_(True)
    out = mod(xyz, xyz, mask, mask, feats)
    (out * torch.from_numpy(fx["probe"]).cuda()).sum().backward()
    got = out.detach().cpu().numpy()
    assert not np.array_equal(got, fx["out"]), "bf16 path produced f32-identical values: it did not run"
    assert _rel_l2(got, fx["out"]) <= BF16_REL_L2 and _rel_max(got, fx["out"]) <= BF16_MAX
    assert _rel_l2(feats.grad.cpu().numpy(), fx["grad_features"]) <= BF16_GRAD_REL_L2
    for k, p in mod.named_parameters():
        if "grad__" + k in fx:
            assert _rel_l2(p.grad.cpu().numpy(), fx["grad__" + k]) <= BF16_GRAD_REL_L2, k


def test_pointwisemlp_bf16_at_the_metric_shape_against_f32_engine():
    """B=4 clouds of the metric shape (N=4096, K=32, C=64): bf16 contraction vs the f32 engine on the same weights."""
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    from oracle import operators as oo
    rng = np.random.default_rng(11)
    B, N, K, C = 4, 4096, 32, 64
    xyz_np, mask_np = oo.make_cloud(rng, B, N, pad_frac=0.1)
    xyz, mask = torch.from_numpy(xyz_np).cuda(), torch.from_numpy(mask_np).cuda()
    f_np = rng.standard_normal((B, C, N)).astype(np.float32)
    probe = torch.from_numpy(rng.standard_normal((B, C, N)).astype(np.float32)).cuda()
    res = {}
    for prec in ("f32", "bf16"):
        torch.manual_seed(5)
        cfg = default_config("pointwisemlp", {"pointwisemlp__feature_type": "dp_fi_df"}, cl3d_impl="fused",
                             cl3d_precision=prec)
        mod = LocalAggregation(C, C, 0.14, K, cfg).cuda().train(True)
        feats = torch.from_numpy(f_np).cuda().requires_grad_(True)
        out = mod(xyz, xyz, mask, mask, feats)
        (out * probe).sum().backward()
        res[prec] = (out.detach().cpu().numpy(), feats.grad.cpu().numpy(),
                     {k: p.grad.cpu().numpy() for k, p in mod.named_parameters()})
    assert _rel_l2(res["bf16"][0], res["f32"][0]) <= BF16_REL_L2
    assert _rel_max(res["bf16"][0], res["f32"][0]) <= BF16_MAX
    assert _rel_l2(res["bf16"][1], res["f32"][1]) <= BF16_GRAD_REL_L2
    for k in res["f32"][2]:
        assert _rel_l2(res["bf16"][2][k], res["f32"][2][k]) <= BF16_GRAD_REL_L2, k


def test_resnet_pointwisemlp_bf16_against_f32_engine():
    """A 5-stage backbone (config 2's structure at a quarter of its size: 8 clouds x 1024 points, width 48, K=16) with
    every contraction (PointWiseMLP rows and 1x1 convolutions) in bf16 against the same network in f32.

    Geometry is bit-identical.  Features are compared STAGE BY STAGE ON IDENTICAL INPUTS (each stage of the bf16
    network is fed the f32 network's input to that stage): that is what the precision of a block changes, and it must
    stay within BF16_NET_REL_L2.  End to end the two randomly initialised networks drift apart much further (measured:
    0.28-0.43 relative L2 at res5) because every max over K and every ReLU is a discontinuity that rounding noise of
    2^-9 flips for near-ties, and five stages of freshly initialised BatchNorm amplify each flip; that number is
    printed, not asserted -- it is a property of the network's conditioning, the same in any mixed-precision run."""
    from closerlook3d_amd.backbones import ResNet
    from closerlook3d_amd.pt_utils import ball_query_cache
    from oracle import operators as oo
    rng = np.random.default_rng(21)
    B, N, K = 8, 1024, 16
    xyz_np, mask_np = oo.make_cloud(rng, B, N, pad_frac=0.05)
    xyz, mask = torch.from_numpy(xyz_np).cuda(), torch.from_numpy(mask_np).cuda()
    feats = xyz.transpose(1, 2).contiguous()
    nets, eps = {}, {}
    for prec in ("f32", "bf16"):
        torch.manual_seed(9)
        cfg = default_config("pointwisemlp", {"pointwisemlp__feature_type": "dp_fi_df"}, cl3d_precision=prec)
        nets[prec] = ResNet(cfg, 3, 0.12, 0.05, [K] * 5, [256, 64, 16, 8], width=48, depth=2, bottleneck_ratio=2).cuda().train(True)
        with ball_query_cache(), torch.no_grad():
            eps[prec] = nets[prec](xyz, mask, feats)
    for stage in range(1, 6):
        assert torch.equal(eps["bf16"][f"res{stage}_xyz"], eps["f32"][f"res{stage}_xyz"])
        assert torch.equal(eps["bf16"][f"res{stage}_mask"], eps["f32"][f"res{stage}_mask"])
    end_to_end = _rel_l2(eps["bf16"]["res5_features"].cpu().numpy(), eps["f32"]["res5_features"].cpu().numpy())
    worst = 0.0
    with torch.no_grad():
        for stage in range(2, 6):  # stage s = layer{s-1}, fed the f32 network's res{s-1} products
            x_in = (eps["f32"][f"res{stage - 1}_xyz"], eps["f32"][f"res{stage - 1}_mask"], eps["f32"][f"res{stage - 1}_features"])
            outs = {}
            for prec in ("f32", "bf16"):
                with ball_query_cache():
                    outs[prec] = getattr(nets[prec], f"layer{stage - 1}")(*x_in)[2]
            assert not torch.equal(outs["bf16"], outs["f32"])
            worst = max(worst, _rel_l2(outs["bf16"].cpu().numpy(), outs["f32"].cpu().numpy()))
    print(f"bf16 backbone: worst stage on identical inputs {worst:.3e} relative L2; end to end at res5 {end_to_end:.3e}")
    assert worst <= BF16_NET_REL_L2


BF16_NET_REL_L2 = 5e-2  # one stage (two bottlenecks: six convolutions, two operators, max-pool) on identical inputs


def test_resnet_with_every_bottleneck_fused_matches_the_grouped_network(monkeypatch):
    """at the fixtures' size (B*N < 16 384) backbones._FUSE_MIN_VALUES routes every bottleneck layer by layer,
    so fused.pointwise_bottleneck only ever ran as a single bottleneck.  Here the threshold is 0: the whole ResNet +
    segmentation head with every PointWiseMLP bottleneck fused, under ball_query_cache() -- the blocks of a stage share
    one idx, one CSR table and one support summary -- against the same network on the grouped dataflow (the reference's
    tensor algebra on the engine's native ops) and against the reference's own fixture."""
    from closerlook3d_amd import backbones
    from closerlook3d_amd.backbones import ResNet, SceneSegHeadResNet
    from closerlook3d_amd.pt_utils import ball_query_cache
    fx = load_fixture("operators_resnet_seg_pointwisemlp.npz")
    K = 16
    res = {}
    for impl in ("auto", "grouped"):
        monkeypatch.setattr(backbones, "_FUSE_MIN_VALUES", 0 if impl == "auto" else 1 << 30)
        cfg = default_config("pointwisemlp", fx["over"], cl3d_impl=impl)
        net = ResNet(cfg, 3, 0.1, 0.05, [K] * 5, [128, 48, 16, 8], width=12, depth=2, bottleneck_ratio=2)
        head = SceneSegHeadResNet(5, 12, 0.1, [K] * 5)
        net.load_state_dict(state_of(fx, "backbone."), strict=True)
        head.load_state_dict(state_of(fx, "head."), strict=True)
        net, head = net.cuda().train(True), head.cuda().train(True)
        feats = torch.from_numpy(fx["features"]).cuda().requires_grad_(True)
        with ball_query_cache():
            ep = net(torch.from_numpy(fx["xyz"]).cuda(), torch.from_numpy(fx["mask"]).cuda(), feats)
            logits = head(ep)
        (logits * torch.from_numpy(fx["probe"]).cuda()).sum().backward()
        res[impl] = (logits.detach().cpu().numpy(), ep["res5_features"].detach().cpu().numpy(), feats.grad.cpu().numpy(),
                     {k: p.grad.cpu().numpy() for k, p in net.named_parameters() if p.grad is not None})
    a, b = res["auto"], res["grouped"]
    assert_close(a[0], fx["out"], 2e-4, "logits vs the reference fixture")
    assert_close(a[0], b[0], 2e-4, "logits fused vs grouped")
    assert_close(a[1], b[1], 2e-4, "res5 fused vs grouped")
    # gradients go through ten layers of arg-max routing: a near-tie resolved the other way moves a whole entry, so the
    # bulk is held tightly and the norm of the difference loosely (as tests/test_fp64_anchor_gpu.py does for the input)
    for name, x, y in [("input", a[2], b[2])] + [(k, a[3][k], b[3][k]) for k in sorted(b[3])]:
        scale = float(np.abs(y).max()) + 1e-12
        moved = float((np.abs(x - y) > 2e-4 * scale + 2e-4 * np.abs(y)).mean())
        assert moved <= 0.08, f"{name}: {moved:.1%} of the gradient entries moved"
        assert np.linalg.norm(x - y) <= 5e-2 * np.linalg.norm(y) + 1e-9, name
    assert set(a[3]) == set(b[3])

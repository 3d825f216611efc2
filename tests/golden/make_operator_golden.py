"""Generate tests/golden/operators_*.npz by running the REFERENCE's own Python modules.

Runs only in the build container (needs /root/reference; it is imported, never copied).  The
reference's five native ops have no CPU implementation, so `pt_custom_ops._ext` is bound to the
oracle's C restatement (oracle.native) -- which is itself pinned against the reference's compiled
kernels on the GPU box (tests/test_ref_pin_gpu.py, tests/golden/native_*.npz).  Everything above
`_ext` -- pt_utils.py, local_aggregation_operators.py, resnet.py, segmentation_head.py -- is the
reference's code, executed as is.

Each fixture stores inputs, every parameter/buffer of the module (state_dict), the forward output
and the gradients of `sum(output * probe)` w.r.t. the input features and all parameters.

    python tests/golden/make_operator_golden.py                       # everything
    python tests/golden/make_operator_golden.py --only bottlenecks    # tests/golden/bottleneck_*.npz alone
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/pytorch"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import operators as oo  # noqa: E402


def _install_stubs():
    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = EasyDict(v) if isinstance(v, dict) and not isinstance(v, EasyDict) else v

        __setitem__ = lambda self, k, v: dict.__setitem__(self, k, EasyDict(v) if isinstance(v, dict) and not isinstance(v, EasyDict) else v)  # noqa: E731

    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed
    pkg = types.ModuleType("pt_custom_ops")
    ext = types.ModuleType("pt_custom_ops._ext")
    for name in ("group_points", "group_points_grad", "masked_ordered_ball_query", "masked_grid_subsampling",
                 "masked_nearest_query"):
        setattr(ext, name, getattr(oo.ExtCPU, name))
    pkg._ext = ext
    sys.modules["pt_custom_ops"] = pkg
    sys.modules["pt_custom_ops._ext"] = ext


def _config(kind, **over):
    from utils.config import config as base
    import copy
    cfg = copy.deepcopy(base)
    cfg.local_aggregation_type = kind
    for k, v in over.items():
        sub, _, leaf = k.partition("__")
        if leaf:
            cfg[sub][leaf] = v
        else:
            cfg[k] = v
    return cfg


def _inputs(seed, B, N, C, pad_frac, kind="uniform"):
    rng = np.random.default_rng(seed)
    xyz, mask = oo.make_cloud(rng, B, N, kind=kind, pad_frac=0.0)
    if pad_frac > 0:  # pad only the last cloud so both padded and full clouds are covered
        x1, m1 = oo.make_cloud(rng, 1, N, kind=kind, pad_frac=pad_frac)
        xyz[-1], mask[-1] = x1[0], m1[0]
    feats = rng.standard_normal((B, C, N)).astype(np.float32)
    return xyz, mask, feats


def _radius(N, K, mult):
    return float((mult * K * 3 / (4 * np.pi * N)) ** (1 / 3))


def _run_module(mod, args, feat_index, seed, max_grad_numel=None):
    torch.manual_seed(seed)
    args = [a.clone() for a in args]
    args[feat_index].requires_grad_(True)
    out = mod(*args)
    outs = out if isinstance(out, (tuple, list)) else (out,)
    y = outs[-1]
    probe = torch.randn(y.shape, generator=torch.Generator().manual_seed(seed + 1))
    (y * probe).sum().backward()
    rec = {"out": y.detach().numpy(), "probe": probe.numpy(), "grad_features": args[feat_index].grad.numpy()}
    for i, o in enumerate(outs[:-1]):
        rec[f"out{i}"] = o.detach().numpy()
    for k, v in mod.named_parameters():
        if v.grad is not None and (max_grad_numel is None or v.numel() <= max_grad_numel):
            rec["grad__" + k] = v.grad.numpy()
    return rec


def _state(mod):
    return {"state__" + k: v.detach().clone().numpy() for k, v in mod.state_dict().items()}


def _randomize_bn(mod, seed):
    g = torch.Generator().manual_seed(seed)
    for m in mod.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            with torch.no_grad():
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))


BOTTLENECK_CASES = [  # name, kind, overrides, (cin, cout), strided
    ("pospool_xyz_plain", "pospool", dict(pospool__position_embedding="xyz", pospool__reduction="avg"), (48, 48), False),
    ("pospool_sincos_strided", "pospool", dict(pospool__position_embedding="sin_cos", pospool__reduction="avg"), (24, 48), True),
    ("adaptive_weight_plain", "adaptive_weight", dict(adaptive_weight__num_mlps=1, adaptive_weight__reduction="avg"), (48, 48), False),
    ("adaptive_weight_strided", "adaptive_weight", dict(adaptive_weight__num_mlps=1, adaptive_weight__reduction="avg"), (24, 48), True),
    ("pseudo_grid_plain", "pseudo_grid", dict(pseudo_grid__KP_influence="linear"), (48, 48), False),
    ("pseudo_grid_strided", "pseudo_grid", dict(pseudo_grid__KP_influence="linear"), (24, 48), True),
]


def bottlenecks(Bottleneck):
    """Whole `Bottleneck`s of the three gather-and-reduce operators (backbones/resnet.py:22-68), plain and strided, every
    parameter gradient kept: the reference pins of fused.reduce_bottleneck.  The strided PosPool-xyz
    one is operators_strided_bottleneck.npz above."""
    B, N, K = 2, 256, 16
    for ci, (name, kind, over, (cin, cout), strided) in enumerate(BOTTLENECK_CASES):
        cfg = _config(kind, **over)
        xyz, mask, feats = _inputs(1000 + ci, B, N, cin, 0.25)
        torch.manual_seed(1100 + ci)
        np.random.seed(1100 + ci)
        btn = Bottleneck(cin, cout, 2, 0.15, K, cfg, downsample=strided, sampleDl=0.12 if strided else None,
                         npoint=64 if strided else None)
        _randomize_bn(btn, 1200 + ci)
        btn.train(True)
        state = _state(btn)
        rec = _run_module(btn, [torch.from_numpy(a) for a in (xyz, mask, feats)], 2, 1300 + ci)
        for k, v in btn.state_dict().items():  # BatchNorm running statistics AFTER the step
            if "running_" in k:
                rec["after__" + k] = v.detach().clone().numpy()
        rec.update(state)
        rec.update(xyz=xyz, mask=mask, features=feats, kind=np.array(kind), over=np.array(repr(over)),
                   cin=np.int32(cin), cout=np.int32(cout), strided=np.int32(strided), radius=np.float32(0.15),
                   nsample=np.int32(K), sampleDl=np.float32(0.12), npoint=np.int32(64))
        np.savez_compressed(os.path.join(OUT, f"bottleneck_{name}.npz"), **rec)
        print("bottleneck", name, rec["out"].shape, float(np.abs(rec["out"]).mean()))


def main(only=None):
    os.environ["JOB_LOG_DIR"] = tempfile.mkdtemp(prefix="cl3d_golden_")
    _install_stubs()
    sys.path.insert(0, REF)
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)
    from models.local_aggregation_operators import LocalAggregation
    from models.backbones.resnet import ResNet, Bottleneck
    from models.heads.segmentation_head import SceneSegHeadResNet
    import pt_utils as ref_pt_utils  # the reference's own file (ops/pt_custom_ops is on sys.path now)
    assert ref_pt_utils.__file__.startswith(REF), ref_pt_utils.__file__

    if only == "bottlenecks":
        bottlenecks(Bottleneck)
        return
    B, N, K, C = 2, 256, 16, 12
    cases = [
        ("pospool_xyz_avg", "pospool", dict(pospool__position_embedding="xyz", pospool__reduction="avg"), 1.5, 0.25, True),
        ("pospool_xyz_max", "pospool", dict(pospool__position_embedding="xyz", pospool__reduction="max"), 1.5, 0.25, True),
        ("pospool_xyz_sum_conv", "pospool", dict(pospool__position_embedding="xyz", pospool__reduction="sum", pospool__output_conv=True), 4.0, 0.0, True),
        ("pospool_sincos_avg", "pospool", dict(pospool__position_embedding="sin_cos", pospool__reduction="avg"), 1.5, 0.25, True),
        ("adaptive_dp_fc1_avg", "adaptive_weight", dict(adaptive_weight__num_mlps=1, adaptive_weight__reduction="avg"), 1.5, 0.25, True),
        ("adaptive_dp_fc2_s2_sum", "adaptive_weight", dict(adaptive_weight__num_mlps=2, adaptive_weight__shared_channels=2, adaptive_weight__reduction="sum"), 4.0, 0.0, True),
        ("pointwisemlp_fc1_max_train", "pointwisemlp", dict(pointwisemlp__feature_type="dp_fi_df", pointwisemlp__num_mlps=1, pointwisemlp__reduction="max"), 1.5, 0.25, True),
        ("pointwisemlp_fc1_max_eval", "pointwisemlp", dict(pointwisemlp__feature_type="dp_fi_df", pointwisemlp__num_mlps=1, pointwisemlp__reduction="max"), 1.5, 0.25, False),
        ("pointwisemlp_fc2_max_train", "pointwisemlp", dict(pointwisemlp__feature_type="dp_fi_df", pointwisemlp__num_mlps=2, pointwisemlp__reduction="max"), 4.0, 0.0, True),
        ("pseudogrid_linear", "pseudo_grid", dict(pseudo_grid__KP_influence="linear"), 1.5, 0.25, True),
        ("pseudogrid_constant_conv", "pseudo_grid", dict(pseudo_grid__KP_influence="constant", pseudo_grid__output_conv=True), 4.0, 0.0, True),
    ]
    for ci, (name, kind, over, mult, pad, train) in enumerate(cases):
        cfg = _config(kind, **over)
        radius = _radius(N, K, mult)
        torch.manual_seed(100 + ci)
        np.random.seed(100 + ci)
        mod = LocalAggregation(C, C, radius, K, cfg)
        _randomize_bn(mod, 200 + ci)
        mod.train(train)
        xyz, mask, feats = _inputs(300 + ci, B, N, C, pad)
        t = [torch.from_numpy(a) for a in (xyz, xyz, mask, mask, feats)]
        state = _state(mod)  # before forward: BN running stats as they enter the step
        rec = _run_module(mod, t, 4, 400 + ci)
        rec.update(state)
        rec.update(xyz=xyz, mask=mask, features=feats, radius=np.float32(radius), nsample=np.int32(K),
                   training=np.int32(train), kind=np.array(kind), over=np.array(repr(over)))
        np.savez_compressed(os.path.join(OUT, f"operators_{name}.npz"), **rec)
        print(name, rec["out"].shape, float(np.abs(rec["out"]).mean()))

    # strided bottleneck (MaskedMaxPool + ball query on barycentres) and a small 5-stage backbone + seg head
    cfg = _config("pospool", pospool__position_embedding="xyz", pospool__reduction="avg")
    xyz, mask, feats = _inputs(900, B, N, 24, 0.25)
    torch.manual_seed(901)
    btn = Bottleneck(24, 48, 2, 0.15, K, cfg, downsample=True, sampleDl=0.12, npoint=64)
    _randomize_bn(btn, 902)
    btn.train(True)
    state = _state(btn)
    rec = _run_module(btn, [torch.from_numpy(a) for a in (xyz, mask, feats)], 2, 903)
    rec.update(state)
    rec.update(xyz=xyz, mask=mask, features=feats)
    np.savez_compressed(os.path.join(OUT, "operators_strided_bottleneck.npz"), **rec)
    print("strided_bottleneck", rec["out"].shape)

    # (small width keeps the fixtures small; parameter gradients are kept for tensors <= 2048 elements)
    for kind, over in (("pospool", dict(pospool__position_embedding="xyz", pospool__reduction="avg")),
                       ("pointwisemlp", dict(pointwisemlp__feature_type="dp_fi_df", pointwisemlp__num_mlps=1, pointwisemlp__reduction="max"))):
        cfg = _config(kind, **over)
        Nb = 512
        xyz, mask, _ = _inputs(950, B, Nb, 3, 0.2)
        feats = np.ascontiguousarray(xyz.transpose(0, 2, 1))
        torch.manual_seed(951)
        np.random.seed(951)
        net = ResNet(cfg, 3, 0.1, 0.05, [K, K, K, K, K], [128, 48, 16, 8], width=12, depth=2, bottleneck_ratio=2)
        head = SceneSegHeadResNet(5, 12, 0.1, [K, K, K, K, K])
        _randomize_bn(net, 952)
        _randomize_bn(head, 953)
        net.train(True)
        head.train(True)

        class Both(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.backbone, self.head = net, head

            def forward(self, xyz, mask, features):
                ep = self.backbone(xyz, mask, features)
                return (ep['res5_xyz'], ep['res5_mask'], ep['res5_features'], ep['res3_xyz'], self.head(ep))

        both = Both()
        state = _state(both)
        rec = _run_module(both, [torch.from_numpy(a) for a in (xyz, mask, feats)], 2, 954, max_grad_numel=2048)
        rec.update(state)
        rec.update(xyz=xyz, mask=mask, features=feats, kind=np.array(kind), over=np.array(repr(over)))
        np.savez_compressed(os.path.join(OUT, f"operators_resnet_seg_{kind}.npz"), **rec)
        print("resnet", kind, rec["out"].shape, float(np.abs(rec["out"]).mean()))


    bottlenecks(Bottleneck)

    # parameter/buffer names and shapes of the remaining caller (no forward needed): checkpoint compatibility
    from models.heads.segmentation_head import MultiPartSegHeadResNet
    import json
    mp = MultiPartSegHeadResNet(3, 12, 0.1, [16] * 5, [4, 2, 6])
    with open(os.path.join(OUT, "state_dict_multipart_head.json"), "w") as fh:
        json.dump({k: list(v.shape) for k, v in mp.state_dict().items()}, fh, indent=0)


if __name__ == "__main__":
    main(only=sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None)

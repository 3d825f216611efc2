"""float64 anchors for the operator fixtures: the REFERENCE's own Python modules, run in double precision.

BASELINE.md section 2 asks for "aggregated features and gradients within 1e-5 (fp32)".  Outputs hold that bound
against the reference's float32 results directly; gradients are sums of thousands of float32 products taken in a
different order, so two correct float32 implementations differ from each other by more than either differs from
the exact value.  This script produces that exact value (to double precision): for every fixture of
make_operator_golden.py it rebuilds the same reference module (LocalAggregation / Bottleneck / ResNet + head), loads
the fixture's parameters, converts it to float64 and runs the fixture's inputs through it.  tests/ then assert that
the engine is as close to the anchor as the reference's float32 run is (tests/test_fp64_anchor_gpu.py; the CPU
oracle is held to the same bound in tests/test_fp64_anchor.py).

Index-producing ops (ball query, grid subsampling, nearest query) are float32 by contract (bit-exact indices) and
stay float32: the stub below casts coordinates to float32 for them (exact: the inputs are float32 values) and
gathers / scatter-adds in the dtype it is given.  Runs only in the build container (needs /root/reference).

    python tests/golden/make_fp64_anchor.py
"""
import ast
import glob
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/pytorch"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import native  # noqa: E402
import make_operator_golden as mog  # noqa: E402  (its stubs and helpers; nothing of the reference is copied)


class Ext64:
    """`pt_custom_ops._ext` for float64 feature tensors: index ops on float32 coordinates, gathers in any dtype."""

    @staticmethod
    def group_points(points, idx):
        p = points.detach().numpy()
        i = idx.numpy().astype(np.int64)
        B, C, _ = p.shape
        out = np.take_along_axis(p[:, :, None, :], np.broadcast_to(i[:, None, :, :], (B, C) + i.shape[1:]).reshape(B, C, 1, -1)
                                 if False else i.reshape(B, 1, 1, -1).repeat(C, 1), axis=3)
        return torch.from_numpy(out.reshape(B, C, i.shape[1], i.shape[2]).copy())

    @staticmethod
    def group_points_grad(grad_out, idx, n):
        g = grad_out.detach().numpy()
        i = idx.numpy().astype(np.int64)
        B, C = g.shape[:2]
        out = np.zeros((B, C, n), g.dtype)
        for b in range(B):
            np.add.at(out[b], (slice(None), i[b].reshape(-1)), g[b].reshape(C, -1))
        return torch.from_numpy(out)

    @staticmethod
    def masked_ordered_ball_query(query_xyz, support_xyz, query_mask, support_mask, radius, nsample):
        i, m = native.masked_ordered_ball_query(query_xyz.numpy().astype(np.float32), support_xyz.numpy().astype(np.float32),
                                                query_mask.numpy(), support_mask.numpy(), radius, nsample)
        return [torch.from_numpy(i), torch.from_numpy(m)]

    @staticmethod
    def masked_grid_subsampling(points, mask, nsamples, sampleDl):
        s, m = native.masked_grid_subsampling(points.numpy().astype(np.float32), mask.numpy(), nsamples, sampleDl)
        return [torch.from_numpy(s).to(points.dtype), torch.from_numpy(m)]

    @staticmethod
    def masked_nearest_query(query_xyz, support_xyz, query_mask, support_mask):
        i, m = native.masked_nearest_query(query_xyz.numpy().astype(np.float32), support_xyz.numpy().astype(np.float32),
                                           query_mask.numpy(), support_mask.numpy())
        return [torch.from_numpy(i), torch.from_numpy(m)]


def _load(name):
    z = np.load(os.path.join(OUT, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def _state(fx, prefix=""):
    return {k[len("state__" + prefix):]: torch.from_numpy(np.array(v)) for k, v in fx.items()
            if k.startswith("state__" + prefix)}


def _run64(mod, args, feat_index, probe):
    args = [a.clone() for a in args]
    args[feat_index].requires_grad_(True)
    out = mod(*args)
    y = (out if not isinstance(out, (tuple, list)) else out[-1])
    (y * torch.from_numpy(probe).double()).sum().backward()
    rec = {"out64": y.detach().numpy(), "grad_features64": args[feat_index].grad.numpy()}
    for k, v in mod.named_parameters():
        if v.grad is not None:
            rec["grad64__" + k] = v.grad.numpy()
    return rec


def main():
    os.environ["JOB_LOG_DIR"] = tempfile.mkdtemp(prefix="cl3d_anchor_")
    mog._install_stubs()
    ext = sys.modules["pt_custom_ops._ext"]
    for name in ("group_points", "group_points_grad", "masked_ordered_ball_query", "masked_grid_subsampling",
                 "masked_nearest_query"):
        setattr(ext, name, getattr(Ext64, name))
    sys.path.insert(0, REF)
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("gloo", rank=0, world_size=1)
    from models.local_aggregation_operators import LocalAggregation
    from models.backbones.resnet import ResNet, Bottleneck
    from models.heads.segmentation_head import SceneSegHeadResNet

    def dbl(a):
        return torch.from_numpy(a).double()

    for path in sorted(glob.glob(os.path.join(OUT, "operators_*.npz"))):
        name = os.path.basename(path)
        fx = _load(name)
        K = 16
        if "resnet" in name:
            kind = str(fx["kind"])
            cfg = mog._config(kind, **ast.literal_eval(str(fx["over"])))
            net = ResNet(cfg, 3, 0.1, 0.05, [K] * 5, [128, 48, 16, 8], width=12, depth=2, bottleneck_ratio=2)
            head = SceneSegHeadResNet(5, 12, 0.1, [K] * 5)

            class Both(torch.nn.Module):
                def __init__(self):
                    super().__init__()
                    self.backbone, self.head = net, head

                def forward(self, xyz, mask, features):
                    ep = self.backbone(xyz, mask, features)
                    return (ep['res5_features'], self.head(ep))

            mod = Both()
            mod.load_state_dict(_state(fx), strict=True)
            mod.double().train(True)
            args = [dbl(fx["xyz"]), torch.from_numpy(fx["mask"]), dbl(fx["features"])]
            feat_index = 2
            args2 = [a.clone() for a in args]
            args2[2].requires_grad_(True)
            res5, logits = mod(*args2)
            (logits * torch.from_numpy(fx["probe"]).double()).sum().backward()
            rec = {"out64": logits.detach().numpy(), "res5_features64": res5.detach().numpy(),
                   "grad_features64": args2[2].grad.numpy()}
            for k, v in mod.named_parameters():
                if v.grad is not None and v.numel() <= 2048:
                    rec["grad64__" + k] = v.grad.numpy()
        elif "bottleneck" in name:
            cfg = mog._config("pospool", pospool__position_embedding="xyz", pospool__reduction="avg")
            mod = Bottleneck(24, 48, 2, 0.15, K, cfg, downsample=True, sampleDl=0.12, npoint=64)
            mod.load_state_dict(_state(fx), strict=True)
            mod.double().train(True)
            rec = _run64(mod, [dbl(fx["xyz"]), torch.from_numpy(fx["mask"]), dbl(fx["features"])], 2, fx["probe"])
        else:
            kind = str(fx["kind"])
            cfg = mog._config(kind, **ast.literal_eval(str(fx["over"])))
            C = fx["features"].shape[1]
            mod = LocalAggregation(C, C, float(fx["radius"]), int(fx["nsample"]), cfg)
            mod.load_state_dict(_state(fx), strict=True)
            mod.double().train(bool(fx["training"]))
            xyz, mask = dbl(fx["xyz"]), torch.from_numpy(fx["mask"])
            rec = _run64(mod, [xyz, xyz, mask, mask, dbl(fx["features"])], 4, fx["probe"])
        # how far the reference's own float32 run is from the anchor
        e_out = float(np.abs(fx["out"].astype(np.float64) - rec["out64"]).max())
        e_g = float(np.abs(fx["grad_features"].astype(np.float64) - rec["grad_features64"]).max())
        print(f"{name}: max |ref32 - anchor| out {e_out:.2e} grad_features {e_g:.2e}")
        np.savez_compressed(os.path.join(OUT, name.replace("operators_", "fp64_anchor_")), **rec)


if __name__ == "__main__":
    main()

"""Bit-level A/B of two builds of the library on one PointWiseMLP training step (outputs, every gradient, BatchNorm
buffers): run once per build (CL3D_LIB selects it) with --out, then --compare a.pt b.pt.
  python scripts/micro/ab_bits.py --out /tmp/a.pt;  CL3D_LIB=.../libcl3d_x.so python scripts/micro/ab_bits.py --out /tmp/b.pt
  python scripts/micro/ab_bits.py --compare /tmp/a.pt /tmp/b.pt"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--compare", nargs=2, default=None)
    args = ap.parse_args()
    if args.compare:
        a, b = (torch.load(p) for p in args.compare)
        bad = [k for k in a if not torch.equal(a[k], b[k])]
        print("tensors:", len(a), "differing:", bad)
        sys.exit(1 if bad else 0)
    from bench import make_config, synth_batch
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    res = {}
    for tag, (B, N, K, C, pad) in {"metric": (16, 4096, 32, 64, 0.0), "padded": (3, 1500, 20, 36, 0.2), "wide": (2, 512, 16, 288, 0.1)}.items():
        dev = torch.device("cuda", 0)
        radius = float((1.5 * K * 3 / (4 * np.pi * N)) ** (1 / 3))
        torch.manual_seed(0)
        la = LocalAggregation(C, C, radius, K, make_config("pointwisemlp", "auto")).to(dev).train(True)
        with torch.no_grad():
            la.local_aggregation_operator.mlps.conv0[1].weight.mul_(torch.where(torch.arange(C, device=dev) % 3 == 0, -1.0, 1.0))  # both signs of gamma
        xyz, mask, feats = (torch.from_numpy(a).to(dev) for a in synth_batch(B, N, C, 11))
        if pad:
            mask[:, int(N * (1 - pad)):] = 0
        feats.requires_grad_(True)
        probe = torch.randn(B, C, N, device=dev)
        for step in range(2):
            la.zero_grad(set_to_none=True)
            feats.grad = None
            out = la(xyz, xyz, mask, mask, feats)
            out.backward(probe)
            res[f"{tag}.{step}.out"] = out.detach().cpu()
            res[f"{tag}.{step}.dfeat"] = feats.grad.cpu()
            for n_, p in la.named_parameters():
                res[f"{tag}.{step}.d{n_}"] = p.grad.cpu()
            for n_, b in la.named_buffers():
                res[f"{tag}.{step}.{n_}"] = b.detach().cpu().clone()
    torch.save(res, args.out)


if __name__ == "__main__":
    main()

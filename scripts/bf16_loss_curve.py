"""Does a network whose dense contractions take bf16 inputs still LEARN like its f32 twin?  (the
bf16 config-2 backbone ends 43 % (relative L2) away from the f32 one at res5 on random weights -- tests/
test_config2_fullsize_gpu.py -- which says how far rounding is amplified through five stages of BatchNorm, not whether
training suffers.)

BASELINE config 2 -- ModelNet40-shaped PointWiseMLP ResNet (16 clouds x 4096 points, width 144, depth 2) + the
reference's classification head (heads/classifier.py:17-54) -- trained for --steps steps of SGD on a synthetic
classification task (8 procedurally generated surface classes under random rotation, anisotropic scale and jitter; a fresh
batch every step, the SAME batches, initial parameters and dropout masks for both precisions), once with
cl3d_precision = 'f32' and once with 'bf16' (--seeds n: n independent batch / dropout streams each).  Reports the loss curves (means over tenths of the run), the accuracy on the
training stream over the last quarter of the run and on 8 held-out batches in eval mode (running statistics).

    python scripts/bf16_loss_curve.py --steps 200 > gpurun_out/bf16_loss_curve.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_config  # noqa: E402

NUM_CLASSES = 8


def _surface(kind, n, rng):
    """n points on a unit-sized surface of class `kind` (before rotation / scale / jitter)."""
    u, v = rng.random(n), rng.random(n)
    if kind == 0:    # sphere
        z = 2 * u - 1
        r = np.sqrt(1 - z * z)
        p = np.stack([r * np.cos(2 * np.pi * v), r * np.sin(2 * np.pi * v), z], 1)
    elif kind == 1:  # cube surface
        p = rng.random((n, 3)) * 2 - 1
        ax = rng.integers(0, 3, n)
        p[np.arange(n), ax] = np.where(rng.random(n) < 0.5, -1.0, 1.0)
    elif kind == 2:  # cylinder (side only)
        p = np.stack([np.cos(2 * np.pi * u), np.sin(2 * np.pi * u), 2 * v - 1], 1)
    elif kind == 3:  # torus
        a, b = 2 * np.pi * u, 2 * np.pi * v
        p = np.stack([(0.7 + 0.3 * np.cos(b)) * np.cos(a), (0.7 + 0.3 * np.cos(b)) * np.sin(a), 0.3 * np.sin(b)], 1)
    elif kind == 4:  # cone
        h = np.sqrt(u)
        p = np.stack([h * np.cos(2 * np.pi * v), h * np.sin(2 * np.pi * v), 1 - 2 * h], 1)
    elif kind == 5:  # two crossing planes
        p = np.stack([2 * u - 1, 2 * v - 1, np.zeros(n)], 1)
        half = rng.random(n) < 0.5
        p[half] = p[half][:, [0, 2, 1]]
    elif kind == 6:  # two small spheres
        z = 2 * u - 1
        r = np.sqrt(1 - z * z)
        p = 0.5 * np.stack([r * np.cos(2 * np.pi * v), r * np.sin(2 * np.pi * v), z], 1)
        p[:, 0] += np.where(rng.random(n) < 0.5, -0.5, 0.5)
    else:            # helix tube
        t = 4 * np.pi * u
        p = np.stack([0.7 * np.cos(t) + 0.12 * np.cos(2 * np.pi * v), 0.7 * np.sin(t) + 0.12 * np.sin(2 * np.pi * v), (t / (2 * np.pi) - 1) * 0.5], 1)
    return p


def make_batch(B, N, seed):
    rng = np.random.default_rng(seed)
    xyz = np.empty((B, N, 3), np.float32)
    labels = rng.integers(0, NUM_CLASSES, B)
    for b in range(B):
        p = _surface(int(labels[b]), N, rng)
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))  # random rotation / reflection
        p = (p * (0.8 + 0.4 * rng.random(3))) @ q.T
        p += 0.01 * rng.standard_normal(p.shape)
        p -= p.min(0)
        xyz[b] = (p / p.max() * 0.9 + 0.05).astype(np.float32)  # inside the unit cube, the scale config 2's radii are set for
    return xyz, labels.astype(np.int64)


def run(precision, args, dev, seed=0):
    from closerlook3d_amd.backbones import ClassifierResNet, ResNet
    B, N, radius, dl, nsamples, npoints, width = 16, 4096, 0.05, 0.02, [32] * 5, [1024, 256, 64, 16], 144
    torch.manual_seed(0)
    cfg = make_config("pointwisemlp", "auto")
    cfg["cl3d_precision"] = precision
    net = ResNet(cfg, 3, radius, dl, nsamples, npoints, width=width, depth=2, bottleneck_ratio=2).to(dev)
    head = ClassifierResNet(NUM_CLASSES, width).to(dev)
    params = list(net.parameters()) + list(head.parameters())
    opt = torch.optim.SGD(params, lr=args.lr, momentum=0.9, weight_decay=1e-4)
    mask = torch.ones(B, N, dtype=torch.int32, device=dev)

    def forward(xyz):
        x = torch.from_numpy(xyz).to(dev)
        return head(net(x, mask, x.transpose(1, 2).contiguous()))

    losses, correct = [], []
    torch.manual_seed(1 + seed)  # the dropout masks of the run
    net.train(True)
    head.train(True)
    t0 = time.time()
    for step in range(args.steps):
        xyz, labels = make_batch(B, N, 1000 + 100000 * seed + step)
        y = torch.from_numpy(labels).to(dev)
        opt.zero_grad(set_to_none=True)
        logits = forward(xyz)
        loss = torch.nn.functional.cross_entropy(logits, y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        correct.append(float((logits.argmax(1) == y).float().mean()))
    torch.cuda.synchronize()
    train_s = time.time() - t0
    net.train(False)
    head.train(False)
    held, hl = [], []
    with torch.no_grad():
        for k in range(8):
            xyz, labels = make_batch(B, N, 900000 + k)
            y = torch.from_numpy(labels).to(dev)
            logits = forward(xyz)
            held.append(float((logits.argmax(1) == y).float().mean()))
            hl.append(float(torch.nn.functional.cross_entropy(logits, y)))
    w = max(20, args.steps // 10)
    q = max(1, args.steps // 4)
    return {"precision": precision, "stream_seed": seed, "loss_by_tenths_of_the_run": [round(float(np.mean(losses[i:i + w])), 4) for i in range(0, args.steps, w)],
            "first_loss": round(losses[0], 4), "last_quarter_loss": round(float(np.mean(losses[-q:])), 4),
            "last_quarter_train_accuracy": round(float(np.mean(correct[-q:])), 4),
            "held_out_accuracy": round(float(np.mean(held)), 4), "held_out_loss": round(float(np.mean(hl)), 4),
            "seconds": round(train_s, 1), "losses": [round(x, 4) for x in losses]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--seeds", type=int, default=1, help="independent batch streams / dropout seeds per precision (same initial parameters): "
                    "the spread between two f32 runs is the noise the f32 - bf16 difference has to be read against")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    out = {"what": "BASELINE config 2 (PointWiseMLP ResNet, 16 x 4096 points, width 144) + classification head, %d SGD steps on 8 synthetic "
                   "surface classes; same batches / initial parameters / dropout masks for both precisions" % args.steps,
           "chance_loss": round(float(np.log(NUM_CLASSES)), 4),
           "runs": [run(p, args, dev, sd) for sd in range(args.seeds) for p in ("f32", "bf16")]}
    q = max(1, args.steps // 4)
    tail = {(r["precision"], r["stream_seed"]): float(np.mean(r["losses"][-q:])) for r in out["runs"]}
    out["last_quarter_loss_f32_minus_bf16_same_stream"] = [round(tail[("f32", sd)] - tail[("bf16", sd)], 4) for sd in range(args.seeds)]
    if args.seeds > 1:
        out["last_quarter_loss_spread_between_streams"] = {p: round(max(tail[(p, sd)] for sd in range(args.seeds)) -
                                                                    min(tail[(p, sd)] for sd in range(args.seeds)), 4) for p in ("f32", "bf16")}
    for r in out["runs"]:
        del r["losses"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()

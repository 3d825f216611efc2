"""Blast radius of sessions 50-53: every fused operator (forward + backward) and the native ops on one stream, bf16 / f32 contractions in a
loop on another stream of the SAME process: how many distinct bit patterns over REPS launches on the same operands?"""
import os, sys, threading, time
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench  # noqa: E402
import bf16_repeat_under_load as g  # noqa: E402
import closerlook3d_amd  # noqa: E402
from closerlook3d_amd import pt_utils  # noqa: E402
from closerlook3d_amd.local_aggregation_operators import LocalAggregation  # noqa: E402

dev = torch.device("cuda:0")
REPS = int(os.environ.get("REPS", "40"))
B, N, K, radius = 16, 4096, 32, 0.14


def bits(t):
    return int(t.detach().contiguous().view(torch.int32).long().sum())


def victims():
    for kind in ("pointwisemlp", "pospool", "adaptive_weight", "pseudo_grid"):
        for C in (72, 144) if kind == "pointwisemlp" else (72,):
            torch.manual_seed(0)
            cfg = bench.make_config(kind, "fused")
            la = LocalAggregation(C, C, radius, K, cfg).to(dev).train(True)
            xyz, mask, feats = bench.synth_batch(B, N, C, 5)
            x = torch.from_numpy(xyz).to(dev); m = torch.from_numpy(mask).to(dev); f = torch.from_numpy(feats).to(dev)
            gout = torch.randn(B, C, N, generator=torch.Generator().manual_seed(2)).to(dev)

            def step(la=la, x=x, m=m, f=f, gout=gout):
                ff = f.clone().requires_grad_(True)
                for p in la.parameters():
                    p.grad = None
                out = la(x, x, m, m, ff)
                out.backward(gout)
                return {"out": out, "d features": ff.grad, "d params": torch.cat([p.grad.reshape(-1) for p in la.parameters() if p.grad is not None])}
            if os.environ.get("BACKWARD_ONLY") != "1":
                yield "%s C=%d" % (kind, C), step
                continue
            # the backward kernels alone: ONE forward pass before the load starts (held in `keep`), only backward() under the load
            ff0 = f.clone().requires_grad_(True)
            out0 = la(x, x, m, m, ff0)
            torch.cuda.synchronize()

            def bstep(la=la, ff0=ff0, out0=out0, gout=gout):
                ff0.grad = None
                for p in la.parameters():
                    p.grad = None
                out0.backward(gout, retain_graph=True)
                return {"d features": ff0.grad, "d params": torch.cat([p.grad.reshape(-1) for p in la.parameters() if p.grad is not None])}
            yield "%s C=%d backward only" % (kind, C), bstep
    xyz, mask, feats = bench.synth_batch(B, N, 64, 5)
    x = torch.from_numpy(xyz).to(dev); m = torch.from_numpy(mask).to(dev); f = torch.from_numpy(feats).to(dev)

    def native():
        idx, im = pt_utils._ball_query(x, x, m, m, radius, K)
        ff = f.clone().requires_grad_(True)
        grouped = pt_utils.grouping_operation(ff, idx)
        grouped.backward(grouped.detach())
        return {"idx": idx, "grouped": grouped, "d features": ff.grad}
    if os.environ.get("BACKWARD_ONLY") != "1":
        yield "ball query + group_points + grad", native


def sweep(tag, prec):
    cs = list(g.cases())[5:10]
    stop = [False]
    sa = torch.cuda.Stream()

    def loader():
        torch.cuda.set_device(0)
        with torch.cuda.stream(sa):
            while not stop[0]:
                for name, k, x, W, dy in cs:
                    g.one(k, x, W, dy, prec)
                sa.synchronize()
    th = None
    if prec is not None:
        th = threading.Thread(target=loader)
        th.start()
        time.sleep(1.0)
    sb = torch.cuda.Stream()
    with torch.cuda.stream(sb):
        for name, step in victims():
            seen = {}
            for _ in range(REPS):
                o = step()
                sb.synchronize()
                for k_, t in o.items():
                    seen.setdefault(k_, set()).add(bits(t))
            print("%-28s %-34s %s" % (tag, name, {k_: len(s) for k_, s in seen.items()}), flush=True)
    stop[0] = True
    if th is not None:
        th.join()


if __name__ == "__main__":
    if os.environ.get("BACKWARD_ONLY") != "1":
        sweep("alone", None)
        sweep("beside f32 contractions", 0)
    sweep("beside bf16 contractions", 1)
    if os.environ.get("BACKWARD_ONLY") == "1":
        sweep("beside bf16 contractions", 1)

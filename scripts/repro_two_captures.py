"""Reproducer for DESIGN 6 / VERDICT r3 item 3b: a step cut into TWO HIP graphs -- graph A = forward + the late stages'
backward, graph B = the early stages' backward -- gave wrong, replay-varying early-stage gradients when both captures
forked work onto the engine's index streams.  One process, one device, no collective: the gradients after
A.replay(); B.replay() are compared with the same two-stage backward launched eagerly (exact same kernels).

    python scripts/repro_two_captures.py [--config partnet_adaptive] [--variant NAME ...]
Variants switch one suspect off at a time (see VARIANTS); every line reports the worst relative difference of the
early / late parameter gradients over `--replays` replays and whether two replays agree with each other.
"""
import argparse
import contextlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench import make_config, synth_batch  # noqa: E402
from bench_backbone import CONFIGS  # noqa: E402

VARIANTS = ["no_async", "async_a_only", "async_b_only", "baseline"]


def run(config, variant, replays):
    import faulthandler
    faulthandler.enable()
    note = lambda *a: print("[repro]", *a, file=sys.stderr, flush=True)
    import closerlook3d_amd
    from closerlook3d_amd import fused, pt_utils
    from closerlook3d_amd.backbones import ResNet
    from closerlook3d_amd.pt_utils import ball_query_cache
    kind, B, N, radius, dl, nsamples, npoints, width = CONFIGS[config]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = make_config(kind, "auto")
    net = ResNet(cfg, 3, radius, dl, nsamples, npoints, width=width, depth=2, bottleneck_ratio=2).to(dev).train(True)
    xyz, mask, _ = synth_batch(B, N, 3, 7)
    scale = 1.0 if N <= 16384 else 4.0
    x = torch.from_numpy((xyz * scale).astype(np.float32)).to(dev)
    m = torch.from_numpy(mask).to(dev)
    feats = x.transpose(1, 2).contiguous()
    params = [(n_, p) for n_, p in net.named_parameters() if p.requires_grad]
    late = [p for n_, p in params if n_.startswith(("layer3.", "layer4."))]
    late_ids = {id(p) for p in late}
    early = [p for _, p in params if id(p) not in late_ids]
    held = {}
    cache = contextlib.nullcontext if variant == "no_bq_cache" else ball_query_cache

    def zero():
        for _, p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            else:
                p.grad.zero_()

    def compute_late():
        zero()
        with cache():
            ep = net(x, m, feats)
        cut = ep["res3_features"]
        torch.autograd.backward([ep["res5_features"].square().mean()], inputs=[cut] + late, retain_graph=True)
        held["cut"], held["seed"] = cut, cut.grad

    def compute_early():
        torch.autograd.backward([held["cut"]], grad_tensors=[held["seed"]], inputs=early)

    def grads():
        torch.cuda.synchronize()
        return (torch.cat([p.grad.reshape(-1).double() for p in early]).clone(),
                torch.cat([p.grad.reshape(-1).double() for p in late]).clone())

    def per_param():
        torch.cuda.synchronize()
        return {n_: p.grad.detach().double().clone() for n_, p in params if id(p) not in late_ids}

    # reference: the same two-stage backward, eager, index streams off
    pt_utils.ASYNC_INDEX = False
    note("eager reference")
    for _ in range(2):
        compute_late()
        note("late done")
        compute_early()
        note("early done")
    ref_e, ref_l = grads()
    ref_pp = per_param()
    note("reference gradients", float(ref_e.norm()), float(ref_l.norm()))
    # BatchNorm running statistics move with every forward; they do not enter training-mode outputs or gradients

    def set_async(on):
        pt_utils.ASYNC_INDEX = 'auto' if on else False

    fused_fork = fused.FORK_GRADS
    if variant == "no_fork_grads":
        fused.FORK_GRADS = False

    def capture(fn, pool, on):
        set_async(on)
        g = torch.cuda.CUDAGraph()
        with closerlook3d_amd.whole_step_capture(False), torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
            fn()
        return g

    a_on = variant not in ("no_async", "async_b_only")
    b_on = variant not in ("no_async", "async_a_only")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        compute_late()
        compute_early()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    note("capture A")
    ga = capture(compute_late, None, a_on)
    note("capture B")
    if variant == "fresh_streams_b":
        pt_utils._INDEX_STREAMS.clear()
    gb = capture(compute_early, None if variant == "separate_pools" else ga.pool(), b_on)
    fused.FORK_GRADS = fused_fork
    worst_e = worst_l = 0.0
    seen = []
    note("replays")
    for _ in range(replays):
        ga.replay()
        if variant == "sync_between":
            torch.cuda.synchronize()
        gb.replay()
        e, l_ = grads()
        worst_e = max(worst_e, float((e - ref_e).abs().max() / ref_e.abs().max()))
        worst_l = max(worst_l, float((l_ - ref_l).abs().max() / ref_l.abs().max()))
        seen.append(float(e.norm()))
    got_pp = per_param()
    bad = sorted(((float((got_pp[k] - ref_pp[k]).abs().max() / (ref_pp[k].abs().max() + 1e-30)), k) for k in ref_pp),
                 reverse=True)
    print(json.dumps({"config": config, "variant": variant, "early_worst_rel": worst_e, "late_worst_rel": worst_l,
                      "early_norms_distinct": len(set(seen)), "replays": replays,
                      "worst_params": [(round(e, 6), k) for e, k in bad[:6]],
                      "exact_params": sum(1 for e, _ in bad if e == 0.0), "n_params": len(bad)}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="partnet_adaptive")
    ap.add_argument("--variant", nargs="*", default=VARIANTS)
    ap.add_argument("--replays", type=int, default=12)
    ap.add_argument("--one", default="")
    a = ap.parse_args()
    if a.one:
        run(a.config, a.one, a.replays)
    else:
        import subprocess
        for v in a.variant:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", a.config, "--one", v, "--replays",
                                str(a.replays)], capture_output=True, text=True, timeout=900)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            print(lines[-1] if lines else json.dumps({"variant": v, "rc": r.returncode, "err": r.stderr[-1500:]}), flush=True)

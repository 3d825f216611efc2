"""Input points/sec through a full 5-stage residual backbone step (forward + backward + SGD), BASELINE.json configs.

    python scripts/bench_backbone.py --config modelnet_pointwisemlp   # config 2: N=4096, K=32, B=16
    python scripts/bench_backbone.py --config s3dis_pseudogrid        # config 3: one 40 960-point scene
    python scripts/bench_backbone.py --config partnet_adaptive        # config 4 (per-GPU share: B=4, N=10 000)
    python scripts/bench_backbone.py --config s3dis_pospool_deep      # config 5 (one 81 920-point scene, width x2)
Synthetic clouds, random-init weights, f32.  Prints one JSON line.
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
from bench import make_config, synth_batch  # noqa: E402

CONFIGS = {
    # kind, B, N, radius, sampleDl, nsamples, npoints, width, impl overrides
    "modelnet_small": ("pospool", 1, 1024, 0.1, 0.04, [16] * 5, [256, 64, 16, 4], 144),
    "modelnet_pointwisemlp": ("pointwisemlp", 16, 4096, 0.05, 0.02, [32] * 5, [1024, 256, 64, 16], 144),
    "s3dis_pseudogrid": ("pseudo_grid", 1, 40960, 0.1, 0.04, [26, 31, 38, 41, 39], [10240, 2560, 640, 160], 144),
    "partnet_adaptive": ("adaptive_weight", 4, 10000, 0.05, 0.02, [23, 38, 42, 40, 36], [5120, 1024, 384, 64], 144),
    "s3dis_pospool_deep": ("pospool", 1, 81920, 0.1, 0.04, [26, 31, 38, 41, 39], [20480, 5120, 1280, 320], 288),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="modelnet_pointwisemlp", choices=sorted(CONFIGS))
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--impl", default="auto")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"],
                    help="arithmetic of the dense contractions (PointWiseMLP rows and 1x1 convolutions); BASELINE config 2 is bf16")
    ap.add_argument("--no-cache", action="store_true", help="disable the per-forward ball-query memo")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python")
    args = ap.parse_args()
    kind, B, N, radius, dl, nsamples, npoints, width = CONFIGS[args.config]
    from closerlook3d_amd.backbones import ResNet
    from closerlook3d_amd.pt_utils import ball_query_cache
    import contextlib
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = make_config(kind, args.impl)
    cfg["cl3d_precision"] = args.precision
    if kind == "pospool" and "deep" in args.config:
        cfg.pospool.position_embedding = "sin_cos"
    net = ResNet(cfg, 3, radius, dl, nsamples, npoints, width=width, depth=2, bottleneck_ratio=2).to(dev).train(True)
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    xyz, mask, _ = synth_batch(B, N, 3, 7)
    scale = 1.0 if N <= 16384 else 4.0  # scenes: metres; objects: unit cube
    xyz = (xyz * scale).astype(np.float32)
    x = torch.from_numpy(xyz).to(dev)
    m = torch.from_numpy(mask).to(dev)
    feats = x.transpose(1, 2).contiguous()

    def step():
        opt.zero_grad(set_to_none=True)
        with (contextlib.nullcontext() if args.no_cache else ball_query_cache()):
            ep = net(x, m, feats)
        ep["res5_features"].square().mean().backward()
        opt.step()

    # the step is hundreds of short kernels: replayed as one HIP graph unless --no-graph (same kernels, same work)
    graph = None
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
        except Exception as e:
            print(f"bench_backbone: HIP graph capture failed ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    run = graph.replay if graph is not None else step
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"config": args.config, "operator": kind, "clouds": B, "points": N, "width": width,
                      "precision": args.precision, "launch": "hip_graph" if graph is not None else "eager",
                      "ms_per_step": round(dt * 1e3, 3), "input_points_per_s": round(B * N / dt, 1),
                      "params_M": round(sum(p.numel() for p in net.parameters()) / 1e6, 2),
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))


if __name__ == "__main__":
    main()

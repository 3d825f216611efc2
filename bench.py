#!/usr/bin/env python
"""bench.py -- points/sec through one LocalAggregation forward+backward (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--operator pointwisemlp|pospool|adaptive_weight|pseudo_grid]

One process per GPU (the driver launches N>1 through torch.distributed.run; backend "nccl" = RCCL).
A step = LocalAggregation fwd + bwd (+ gradient all-reduce over RCCL when N>1 + SGD update of the
operator's parameters) over one batch of B=16 synthetic clouds per GPU, N=M=4096 points, K=32
neighbours, C=64 channels (72 for PosPool, which needs C%3==0).  Clouds are resident in HBM before
the timed region.  Prints ONE JSON line on rank 0 with the contract fields plus
  roofline      dominant hand-written kernel of the reference-visible ball_query+group path:
                algorithmic bytes per launch / average launch duration (HIP events on the launch
                stream) against the 8 TB/s HBM peak; `per_kernel` has the same for every kernel
                of that path, `ball_query_group` the whole boundary at SURVEY 8(d)'s 17,696 B/point
  cpu_baseline  the CPU oracle (C restatement of the native ops + torch CPU operators) on a bounded
                sample of the same workload, rank 0, N=1 only.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md); 6.29e12 is the measured float4-copy ceiling
HBM_MEASURED = 6.29e12


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def make_config(kind, impl):
    return Cfg(bn_momentum=0.1, density_parameter=5.0, local_aggregation_type=kind, cl3d_impl=impl,
               pospool=Cfg(position_embedding='xyz', reduction='avg', output_conv=False),
               adaptive_weight=Cfg(weight_type='dp', num_mlps=1, shared_channels=1, weight_softmax=False,
                                   reduction='avg', output_conv=False),
               pointwisemlp=Cfg(feature_type='dp_fi_df', num_mlps=1, reduction='max'),
               pseudo_grid=Cfg(fixed_kernel_points='center', KP_influence='linear', KP_extent=1.0,
                               num_kernel_points=15, convolution_mode='sum', output_conv=False))


def synth_batch(B, N, C, seed, pad_frac=0.0):
    rng = np.random.default_rng(seed)
    xyz = rng.random((B, N, 3), dtype=np.float32)
    mask = np.ones((B, N), np.int32)
    if pad_frac > 0:
        nv = int(N * (1 - pad_frac))
        xyz[:, nv:] = xyz[:, np.arange(nv, N) % nv]
        mask[:, nv:] = 0
    feats = rng.standard_normal((B, C, N)).astype(np.float32)
    return xyz, mask, feats


def event_time_ms(fn, iters, warmup=3, burst=8):
    """Average duration of one fn() in ms: HIP events on the current (= launch) stream around bursts of
    `burst` back-to-back launches, so the device stays busy and host launch latency is not counted."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    rounds = max(1, iters // burst)
    start = [torch.cuda.Event(enable_timing=True) for _ in range(rounds)]
    stop = [torch.cuda.Event(enable_timing=True) for _ in range(rounds)]
    for i in range(rounds):
        start[i].record()
        for _ in range(burst):
            fn()
        stop[i].record()
    torch.cuda.synchronize()
    return float(np.mean([s.elapsed_time(e) for s, e in zip(start, stop)])) / burst


def kernel_rooflines(xyz, mask, feats, radius, K, iters):
    """Per-kernel achieved bandwidth of the materialising ball_query+group path."""
    from closerlook3d_amd import _ext
    B, N, _ = xyz.shape
    C = feats.shape[1]
    M = N
    idx, _ = _ext.masked_ordered_ball_query(xyz, xyz, mask, mask, radius, K)
    grad_out = torch.randn(B, C, M, K, device=xyz.device)
    MK = M * K
    specs = {
        # name: (callable, algorithmic bytes per launch)
        "ball_query_kernel": (lambda: _ext.masked_ordered_ball_query(xyz, xyz, mask, mask, radius, K),
                              B * (12 * M + 12 * N + 4 * M + 4 * N + 8 * MK)),
        "group_fwd_lds_kernel": (lambda: _ext.group_points(feats, idx), B * (4 * C * MK + 4 * C * N + 4 * MK)),
        "group_rel_kernel+group_fwd_lds_kernel": (lambda: _ext.group_xyz_features(xyz, xyz, feats, idx, radius, True),
                                                  B * (12 * M + 12 * N + 4 * C * N + 4 * MK + 12 * MK + 4 * C * MK)),
        "group_bwd_lds_kernel": (lambda: _ext.group_points_grad(grad_out, idx, N), B * (4 * C * MK + 4 * MK + 4 * C * N)),
    }
    out = {}
    for name, (fn, nbytes) in specs.items():
        ms = event_time_ms(fn, iters)
        out[name] = {"ms": round(ms, 5), "bytes": int(nbytes), "achieved_GBps": round(nbytes / ms / 1e6, 1),
                     "frac": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4)}
    # the whole reference-visible boundary: fwd (query + group) and bwd (scatter), 17,696 B/point at the metric shape
    total_ms = out["ball_query_kernel"]["ms"] + out["group_rel_kernel+group_fwd_lds_kernel"]["ms"] + out["group_bwd_lds_kernel"]["ms"]
    fwd = 12 * M + 12 * N + 4 * M + 4 * N + 4 * C * N + 4 * MK + 4 * MK + 12 * MK + 4 * C * MK
    bwd = 4 * C * MK + 4 * MK + 4 * C * N
    boundary = {"ms": round(total_ms, 5), "bytes_per_point": (fwd + bwd) / M,
                "points_per_s": round(B * M / (total_ms * 1e-3), 1),
                "achieved_GBps": round(B * (fwd + bwd) / total_ms / 1e6, 1),
                "frac": round(B * (fwd + bwd) / (total_ms * 1e-3) / HBM_PEAK, 4),
                "frac_of_measured_copy_peak": round(B * (fwd + bwd) / (total_ms * 1e-3) / HBM_MEASURED, 4)}
    return out, boundary


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (scripts/pmc_kernels.py; rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 FETCH_SIZE correction applied), or None."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json"))):
        try:
            data = json.load(open(path))["kernels"]
        except Exception:
            continue
        for name, rec in data.items():
            if name.replace("cl3d::", "").startswith(kernel):
                best = rec["hbm_bytes"]
    return best


def cpu_baseline(kind, N, K, C, radius, clouds, iters):
    """The oracle (port of the reference semantics) timed on the host: LA fwd+bwd, `clouds` clouds."""
    from oracle import native as on  # noqa: F401  (test/bench infrastructure only)
    from oracle import operators as oo
    xyz, mask, feats = synth_batch(clouds, N, C, 12345)
    t = [torch.from_numpy(a) for a in (xyz, xyz, mask, mask)]
    f = torch.from_numpy(feats).requires_grad_(True)
    g = torch.Generator().manual_seed(0)
    if kind == "pointwisemlp":
        W = (torch.randn(C, 3 + 2 * C, generator=g) * 0.05).requires_grad_(True)
        layers = [dict(weight=W, gamma=torch.ones(C, requires_grad=True), beta=torch.zeros(C, requires_grad=True))]
        fn = lambda: oo.pointwise_mlp(*t, f, radius, K, layers, reduction='max', training=True)  # noqa: E731
    elif kind == "pospool":
        fn = lambda: oo.pospool(*t, f, radius, K, 'xyz', 'avg')  # noqa: E731
    elif kind == "adaptive_weight":
        W = (torch.randn(C, 3, generator=g)).requires_grad_(True)
        b = torch.zeros(C, requires_grad=True)
        fn = lambda: oo.adaptive_weight(*t, f, radius, K, [W], [b], 1, 'avg')  # noqa: E731
    else:
        kp = torch.randn(15, 3, generator=g) * 0.05
        kw = (torch.randn(15, C, generator=g) * 0.1).requires_grad_(True)
        fn = lambda: oo.pseudo_grid(*t, f, radius, K, kp, kw, 2 * radius / 5.0, 'linear')  # noqa: E731
    fn().sum().backward()  # warm-up
    ts = []
    for _ in range(iters):
        f.grad = None
        t0 = time.perf_counter()
        fn().sum().backward()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    return {"value": round(clouds * N / med, 1), "unit": "points/s", "cores": int(torch.get_num_threads()),
            "kind": "port",
            "sample": f"{clouds} clouds x {iters} timed fwd+bwd of the same operator/shape (N={N},K={K},C={C}); "
                      f"C oracle for the native ops (1 thread) + torch CPU ops ({torch.get_num_threads()} threads); "
                      f"median {med * 1e3:.1f} ms; host has {os.cpu_count()} logical cores"}


# ---- cpu_baseline legs of the auxiliary benches (scripts/bench_dataset_grid.py, scripts/bench_voting.py): like
# cpu_baseline() above, the only places outside tests/ where the oracle / the reference build is timed
def cpu_baseline_dataset_grid(points, features, labels, dl):
    """The reference's own grid_subsampling.cpp (oracle/_ref/libgrid_dataset_ref.so) on one host thread:
    (seconds, voxels), or None when the library has not been built."""
    from oracle import build_ref
    try:
        lib = build_ref.load_grid()
    except OSError:
        return None
    n = points.shape[0]
    sp = np.empty((n, 3), np.float32)
    sf = np.empty((n, features.shape[1]), np.float32)
    sl = np.empty((n, labels.shape[1]), np.int32)
    t0 = time.perf_counter()
    m = lib.cl3d_ref_dataset_grid_subsampling(points.ctypes.data, features.ctypes.data, labels.ctypes.data, n,
                                              features.shape[1], labels.shape[1], dl, sp.ctypes.data, sf.ctypes.data,
                                              sl.ctypes.data)
    return time.perf_counter() - t0, int(m)


def cpu_baseline_voting(batches, num_classes, cloud_sizes):
    """The reference's per-element host loop (restated in oracle/voting.py) over `batches`: seconds per batch."""
    from oracle import voting as ov
    arrays = ov.new_arrays(num_classes, cloud_sizes)
    t0 = time.perf_counter()
    for pred, mask, inds, label in batches:
        ov.collect(arrays, pred, mask, inds, label)
    return (time.perf_counter() - t0) / len(batches)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--operator", default="pointwisemlp", choices=["pointwisemlp", "pospool", "adaptive_weight", "pseudo_grid"])
    ap.add_argument("--impl", default="auto", choices=["auto", "fused", "grouped"])
    ap.add_argument("--batch", type=int, default=16, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--nsample", type=int, default=32)
    ap.add_argument("--channels", type=int, default=0, help="0 = 64 (72 for pospool)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a HIP graph")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("--gpus N>1 must be launched through torch.distributed.run (one process per GPU)")
    # CL3D_BENCH_ONE_DEVICE=1: every rank on GPU 0 over gloo -- only to exercise the N>1 code path on a 1-GPU box
    one_dev = os.environ.get("CL3D_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from closerlook3d_amd.dp import FlatGradients
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation

    kind = args.operator
    B, N, K = args.batch, args.points, args.nsample
    C = args.channels or (72 if kind == "pospool" else 64)
    radius = float((1.5 * K * 3 / (4 * np.pi * N)) ** (1 / 3))  # mean in-radius count ~1.5K
    if kind == "pseudo_grid":
        radius = float((1.5 * K * 3 / (4 * np.pi * N)) ** (1 / 3))
    torch.manual_seed(0)  # same parameters on every rank
    module = LocalAggregation(C, C, radius, K, make_config(kind, args.impl)).to(dev).train(True)
    params = [p for p in module.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3) if params else None
    xyz, mask, feats = (torch.from_numpy(a).to(dev) for a in synth_batch(B, N, C, 1000 + rank))
    feats.requires_grad_(True)
    probe = torch.randn(B, C, N, device=dev)

    # N > 1: the parameter gradients live in one flat buffer (zeroed inside the captured step, exchanged with a
    # single in-place RCCL all-reduce) and the update is a second, tiny graph -- three host calls per step
    flat = FlatGradients(params) if (world > 1 and params) else None

    def compute():  # forward + backward (+ the parameter update when there is no gradient exchange)
        feats.grad = None
        if flat is not None:
            flat.zero_()
        elif opt is not None:
            opt.zero_grad(set_to_none=True)
        out = module(xyz, xyz, mask, mask, feats)
        out.backward(probe)  # upstream gradient handed in directly: nothing but the operator is timed
        if world == 1 and opt is not None:
            opt.step()

    def capture(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # N > 1: RCCL's watchdog thread polls events while this thread captures; thread-local capture mode keeps
        # another thread's event query from invalidating the capture (this thread makes no unsafe call itself)
        with torch.cuda.graph(g, capture_error_mode="thread_local" if world > 1 else "global"):
            fn()
        return g

    # A step is ~60 short kernels: launched one by one from Python the host, not the GPU, sets the pace
    # (measured: 0.65 ms of kernels in a 0.81 ms step).  So the whole step is captured once into a HIP graph
    # and replayed -- same kernels, same work, one launch.  The RCCL gradient exchange stays outside the graphs.
    graph = update_graph = None
    if not args.no_graph:
        try:
            graph = capture(compute)
            if world > 1 and opt is not None:
                update_graph = capture(opt.step)
        except Exception as e:  # capture not possible: launch eagerly, say so in the JSON line
            print(f"bench: HIP graph capture failed ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
            graph = update_graph = None
            torch.cuda.synchronize()

    def step():
        if graph is not None:
            graph.replay()
        else:
            compute()
        if world > 1:
            if flat is not None:
                flat.allreduce_mean(world)
            if update_graph is not None:
                update_graph.replay()
            elif opt is not None:
                opt.step()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * N / (elapsed / args.steps)
        line = {
            "metric": "points/sec local-aggregation fwd+bwd (N=4096,K=32,C=64)", "value": round(value, 1),
            "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ModelNet40-shape {kind} LocalAggregation fwd+bwd", "operator": kind,
                       "impl": args.impl, "clouds_per_gpu": B, "points": N, "nsample": K, "channels": C,
                       "radius": round(radius, 5), "launch": "hip_graph" if graph is not None else "eager",
                       "parallelism": f"dp{world} (clouds sharded, RCCL grad all-reduce)"},
        }
        if not args.no_kernel_roofline:
            with torch.no_grad():
                per_kernel, boundary = kernel_rooflines(xyz, mask, feats.detach(), radius, K, max(10, args.steps))
            dom = max((k for k in per_kernel if "+" not in k), key=lambda k: per_kernel[k]["ms"])
            line["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": per_kernel[dom]["achieved_GBps"],
                                "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": per_kernel[dom]["frac"],
                                "traffic": pmc_traffic(dom), "per_kernel": per_kernel, "ball_query_group": boundary}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(kind, N, K, C, radius, clouds=4, iters=5)
            line["speedup_vs_cpu_baseline"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()  # rank 0 measured the per-kernel rooflines after the timed region: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Input points/sec through a full 5-stage residual backbone step (forward + backward + SGD), BASELINE.json configs.

    python scripts/bench_backbone.py --config modelnet_pointwisemlp   # config 2: N=4096, K=32, B=16
    python scripts/bench_backbone.py --config s3dis_pseudogrid        # config 3: one 40 960-point scene
    python scripts/bench_backbone.py --config partnet_adaptive        # config 4 (per-GPU share: B=4, N=10 000)
    python scripts/bench_backbone.py --config s3dis_pospool_deep      # config 5 (one 81 920-point scene, width x2)
Synthetic clouds, random-init weights; f32 or, with --precision bf16, bf16 contractions (BASELINE config 2's dtype).
Prints one JSON line on rank 0.

Data parallel (reference function/train_modelnet_dist.py:117-125,206,280): one process per GPU, every rank
the per-GPU share above (weak scaling), parameter gradients in one flat buffer averaged by a single RCCL all-reduce per
step (closerlook3d_amd/dp.py), BatchNorm statistics per rank as in the reference:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        scripts/bench_backbone.py --gpus 8 --config partnet_adaptive
The compute (zero gradients, forward, backward) is one HIP graph, the all-reduce stays outside it, the optimiser step
is a second graph; the line reports the step time (max over ranks), the all-reduce time (HIP events) and its bytes.
CL3D_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 over gloo (exercises the N>1 code path on a 1-GPU box).

Probes (N > 1; how round 6 traced a checksum that would not repeat down to one instruction):
    --checksums                    norms of the last step's gradients, of the parameters, and per parameter
    --repeat-check N               replay the step N times WITHOUT the update: distinct bit patterns of the gradients, by parameter
      ... --no-graph --dump-forward x   also: the first module / autograd Function whose OUTPUT varies over the replays
      ... CL3D_TRACE_PWMLP=1            also: the gather pass's operands and products, read back through hipMemcpy, element by element
    CL3D_DP_DEBUG=1 (norms per warm-up step), CL3D_DP_SYNC=1 (device-wide waits around the exchange),
    CL3D_DP_NOEXCHANGE=1 (with --repeat-check: this rank's own gradients)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

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
    import faulthandler
    import signal
    faulthandler.register(signal.SIGUSR1, all_threads=True)  # `kill -USR1 <rank pid>` prints every thread's Python stack
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="modelnet_pointwisemlp", choices=sorted(CONFIGS))
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--impl", default="auto")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"],
                    help="arithmetic of the dense contractions (PointWiseMLP rows and 1x1 convolutions); BASELINE config 2 is bf16")
    ap.add_argument("--gemm-plans", default="measured", choices=["model", "measured"],
                    help="tile / K-slice plans of the dense products: the launch-time model, or timed at first sight during the "
                         "warm-up steps (closerlook3d_amd.gemm_autotune)")
    ap.add_argument("--maxpool", default="targets", choices=["targets", "slots"],
                    help="max pooling's backward: a scatter on kept support indices (default) or the ordered gather through the CSR "
                         "inverse of rounds 1-5 (A/B arm)")
    ap.add_argument("--no-cache", action="store_true", help="disable the per-forward ball-query memo")
    ap.add_argument("--head", action="store_true",
                    help="backbone + the scene-segmentation head (nearest up-sampling decoder + classifier) in the step")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python")
    ap.add_argument("--self-check", action="store_true", help="compare the two-stage backward with the plain one and exit")
    ap.add_argument("--checksums", action="store_true", help="add the L2 norms of the last step's gradients and of the parameters")
    ap.add_argument("--overlap", action="store_true",
                    help="N > 1: the step as two graphs with the late-stage gradients exchanged while the early stages' "
                         "backward replays, instead of one flat all-reduce after the whole backward")
    ap.add_argument("--overlap-forks", default="", choices=["", "a", "b", "both", "none"],
                    help="--overlap: which of the two graphs keeps the side-stream forks (default a; b / both = the "
                         "configuration that gives wrong early-stage gradients, kept to reproduce it)")
    ap.add_argument("--unsafe", action="store_true",
                    help="required by --overlap-forks b|both WITH --debug-two-graphs other_stream: forked gradient products inside the graph that holds the early "
                         "stages' backward ALONE give wrong, replay-varying gradients (DESIGN 6; kept to reproduce it)")
    ap.add_argument("--debug-two-graphs", default="", help="two-graph reproducer, comma list of: sync_between (device sync between the replays "
                    "of graph A and graph B), own_pool (graph B in a memory pool of its own), fresh_streams (graph B forks "
                    "onto streams graph A never saw), other_stream (the warm-up on a stream of its own, as before round 6: autograd's "
                    "AccumulateGrad nodes run on the stream that was current when a parameter was first used -- the warm-up's -- "
                    "which is then a third branch of every captured backward pass; with --overlap-forks b this is the layout "
                    "that gives replay-varying gradients)")
    ap.add_argument("--fork-mode", default="reuse", choices=["reuse", "serial_side", "after", "probe"],
                    help="two-graph reproducer: how a forked pair of gradient products is laid out (reuse = shipped)")
    ap.add_argument("--fork-only", default="", help="debug: comma list of fork episodes (1-based, counted from graph B's capture) that fork")
    ap.add_argument("--dump-grads", default="", help="rank 0: after the FIRST step save {parameter name: gradient} here and exit")
    ap.add_argument("--repeat-check", type=int, default=0, help="N > 1 ranks: replay the step this many times WITHOUT the "
                    "parameter update (same parameters, same clouds: the same gradients every time) and count the distinct "
                    "bit patterns of the exchanged gradient buffer, late and early part apart")
    ap.add_argument("--comm-on-main", action="store_true", help="--overlap reproducer: the first bucket's exchange on the "
                    "step's own stream (graph B starts behind it: no overlap)")
    ap.add_argument("--dump-forward", default="", help="rank 0, with --no-graph: bit checksums of every module's output in the "
                    "first step, in call order, as JSON (run-to-run determinism probe)")
    ap.add_argument("--lead-kernel", action="store_true",
                    help="--overlap: graph B starts with a trivial kernel on the capture stream, so that no forked branch "
                         "is a ROOT of the graph")
    ap.add_argument("--block", default="engine", choices=["engine", "modules"],
                    help="A/B: 'modules' runs the bottlenecks' convolutions / BatchNorms as nn modules (the round-1 path)")
    ap.add_argument("--decode", default="split", choices=["split", "cat"],
                    help="A/B: 'cat' lets the decoders concatenate as the reference does")
    ap.add_argument("--layerwise", action="store_true",
                    help="A/B: PointWiseMLP bottlenecks layer by layer (the activated tensors between their layers materialised)")
    ap.add_argument("--weight-grads", default="deferred", choices=["joined", "deferred"],
                    help="deferred (default): the contractions' weight gradients stay on the side stream beside the rest of the "
                         "backward pass and are joined once, in front of the optimizer / the gradient exchange "
                         "(closerlook3d_amd.deferred_weight_gradients; config 2 bf16 5.60 -> 5.09 ms, profiles/r06/session35_summary.txt); "
                         "joined: layer by layer, as rounds 3-6 had it.  The two-graph --overlap step always joins layer by layer")
    ap.add_argument("--fork-min-points", type=int, default=0,
                    help="A/B: contractions over fewer points (B * N) than this run their two gradient products in line instead of "
                         "as two branches of the graph (fused.FORK_MIN_POINTS)")
    args = ap.parse_args()
    if args.overlap_forks in ("b", "both") and "other_stream" in args.debug_two_graphs.split(",") and not args.unsafe:
        raise SystemExit("--overlap-forks %s together with --debug-two-graphs other_stream is the KNOWN-BAD layout: a forked pair of "
                         "gradient products inside graph B (the early stages' backward alone) WHILE autograd's AccumulateGrad nodes "
                         "run on the warm-up's stream -- a third concurrent branch of that graph -- gives wrong, replay-varying "
                         "early-stage gradients (38-114 + singletons of 200 replays, profiles/r06/two_graph_repeat_check.txt; "
                         "with warm-up and capture on ONE stream, the default since round 6, the same forks are exact over "
                         "5 x 1000 replays: DESIGN 6).  Add --unsafe to run it anyway, e.g. with --repeat-check 200."
                         % args.overlap_forks)
    kind, B, N, radius, dl, nsamples, npoints, width = CONFIGS[args.config]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # bare shell: become the N-rank launch of this command
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from closerlook3d_amd.dp import self_launch
        self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU")
    one_dev = os.environ.get("CL3D_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    from closerlook3d_amd.dp import prepare_environment
    prepare_environment()  # (before the first torch.cuda call: what RCCL needs from the HIP runtime on this driver)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if one_dev else "nccl", **({} if one_dev else {"device_id": dev}))
        from closerlook3d_amd.dp import device_identity, rank_census
        rank_census(device_identity(dev))  # raises under RCCL when two ranks share a device
    import closerlook3d_amd
    closerlook3d_amd.gemm_autotune(args.gemm_plans == "measured")
    if args.block != "engine" or args.decode != "split" or args.layerwise:  # A/B arms, installed from outside the package
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from ab import library_arms
        library_arms.install(block=args.block, decode=args.decode, layerwise=args.layerwise)
    from closerlook3d_amd import fused as _fu
    _fu.FORK_MIN_POINTS = args.fork_min_points
    from closerlook3d_amd import pt_utils as _put
    _fu.MAXPOOL_TARGETS = args.maxpool == "targets"
    _fork_shipped, _fork_count = _fu._fork_join, [0]

    def _fork_debug(device, side_fn, main_fn, points=None, defer=None):
        """The engine's fork / join of two gradient products with the reproducer's switches: only some
        episodes of a capture fork; the two pieces one behind the other on the same two streams instead of side by side."""
        if not (device.type == 'cuda' and _put.async_index() and _fu._forks_allowed()):
            return _fork_shipped(device, side_fn, main_fn, points, defer)
        _fork_count[0] += 1
        if fork_only is not None and _fork_count[0] not in fork_only:
            side_fn()
            main_fn()
            return
        main, side = torch.cuda.current_stream(device), _put.index_stream(device, 2)
        if args.fork_mode == "serial_side":  # both pieces on the side stream, one after the other
            side.wait_stream(main)
            with torch.cuda.stream(side):
                side_fn()
                main_fn()
            main.wait_stream(side)
        elif args.fork_mode == "after":  # the side piece forked behind the caller's piece: never concurrent
            main_fn()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                side_fn()
            main.wait_stream(side)
        elif args.fork_mode == "probe":
            # ordering probe: the caller's stream counts the episode before the fork, the side stream counts it after its
            # piece and compares -- equal whenever the fork edge (side after the caller's count) and the join edge (the
            # caller's next count after the side's check) hold; the same again from the caller's side after the join
            pr = _probe
            if "c_main" not in pr:
                raise RuntimeError("probe buffers must exist before the capture")
            pr["c_main"].add_(1)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pr["early"].add_((pr["c_side"] + 1 != pr["c_main"]).long())  # side started before the caller's count?
                side_fn()
                pr["c_side"].add_(1)
                ev = torch.cuda.Event()
                ev.record(side)
            main_fn()
            main.wait_event(ev)
            pr["late"].add_((pr["c_side"] != pr["c_main"]).long())  # the caller went on before the side's piece ended?
        else:
            return _fork_shipped(device, side_fn, main_fn, points, defer)

    _probe = {}
    if args.fork_mode == "probe":
        for k_ in ("c_main", "c_side", "early", "late"):
            _probe[k_] = torch.zeros(1, dtype=torch.int64, device=dev)
    fork_only = {int(t) for t in args.fork_only.split(",")} if args.fork_only else None
    if fork_only is not None or args.fork_mode != "reuse":
        _fu._fork_join = _fork_debug
    from closerlook3d_amd.backbones import ResNet
    from closerlook3d_amd.dp import FlatGradients
    from closerlook3d_amd.pt_utils import ball_query_cache
    import contextlib
    torch.manual_seed(0)  # same parameters on every rank
    cfg = make_config(kind, args.impl)
    cfg["cl3d_precision"] = args.precision
    if kind == "pospool" and "deep" in args.config:
        cfg.pospool.position_embedding = "sin_cos"
    net = ResNet(cfg, 3, radius, dl, nsamples, npoints, width=width, depth=2, bottleneck_ratio=2).to(dev).train(True)
    head = None
    if args.head:  # the scene-segmentation decoder + classifier of the reference's S3DIS / PartNet models (13 classes)
        from closerlook3d_amd.backbones import SceneSegHeadResNet
        head = SceneSegHeadResNet(13, width, radius, nsamples, config=cfg).to(dev).train(True)
    params = [p for p in net.parameters() if p.requires_grad] + ([p for p in head.parameters()] if head is not None else [])
    opt = torch.optim.SGD(params, lr=1e-3)
    xyz, mask, _ = synth_batch(B, N, 3, 7 + rank)  # every rank its own clouds / scene
    scale = 1.0 if N <= 16384 else 4.0  # scenes: metres; objects: unit cube
    xyz = (xyz * scale).astype(np.float32)
    x = torch.from_numpy(xyz).to(dev)
    m = torch.from_numpy(mask).to(dev)
    feats = x.transpose(1, 2).contiguous()
    # N > 1: the step is cut where most parameters sit behind most of the backward TIME.  The two widest stages
    # (layer3, layer4, the head) hold ~94 % of the parameters, the early stages the points: graph A = zero gradients +
    # forward + backward down to layer3's input; the late-stage gradients (the head of the flat buffer) are then
    # all-reduced on RCCL's stream WHILE graph B replays the backward of the early stages; the small remainder follows.
    # (reference: DistributedDataParallel's bucketed overlap, function/train_modelnet_dist.py:206,280)
    overlap = world > 1 and args.overlap
    if overlap:
        # Graph A (forward + late-stage backward) keeps the engine's side-stream forks; graph B (the early stages'
        # backward alone) is captured single-stream.  Round 3 had found wrong, replay-varying early-stage gradients with
        # forks in both graphs; round 4 narrowed it: forks in A only are exact, ONE forked weight / data
        # gradient pair inside graph B -- the third, layer2's strided conv2 in the small config -- is enough to corrupt
        # results that graph B computed BEFORE that fork, only when the two products run concurrently (the same launches
        # one behind the other on the same two streams are exact), independent of memory pool, stream reuse and of a
        # device sync between the two replays.  --overlap-forks b|both re-creates it.
        from closerlook3d_amd import pt_utils as _pu
        _pu.ASYNC_INDEX = False  # (set per capture below)
        if not args.overlap_forks:
            args.overlap_forks = "a"
    late = [p for n_, p in net.named_parameters() if p.requires_grad and n_.startswith(("layer3.", "layer4."))]
    late += [p for p in head.parameters()] if head is not None else []
    late_ids = {id(p) for p in late}
    early = [p for p in params if id(p) not in late_ids]
    flat = FlatGradients(late + early) if world > 1 else None
    n_late = sum(p.numel() for p in late)
    # the cut: layer3's input.  The segmentation head also reads res1 .. res3 through its skip connections; it is fed
    # DETACHED copies of them, so that graph A's backward stops there (a plain `inputs=[res1, res2, res3]` would walk
    # layer1 and layer2 to form the total derivatives) and graph B starts from [res3, res2, res1] with the skip
    # gradients as seeds -- autograd adds what flows down the chain.
    skip_keys = [] if head is None else ["res1_features", "res2_features", "res3_features"]
    held = {}

    def compute_late():  # graph A
        flat.zero_()
        with (contextlib.nullcontext() if args.no_cache else ball_query_cache()):
            ep = net(x, m, feats)
            det = {k: ep[k].detach().requires_grad_(True) for k in skip_keys}
            out = head({**ep, **det}) if head is not None else ep["res5_features"]
        main_cut = ep["res3_features"]
        torch.autograd.backward([out.square().mean()], inputs=[main_cut] + list(det.values()) + late, retain_graph=True)
        seeds = {"res3_features": main_cut.grad}
        for k, d in det.items():
            seeds[k] = d.grad if k not in seeds else seeds[k] + d.grad
        held["cuts"] = [ep[k] for k in seeds]   # (kept alive: graph B walks the graph below them)
        held["seeds"] = [seeds[k] for k in seeds]

    def compute_early():  # graph B
        seeds = held["seeds"]
        if args.lead_kernel:
            seeds = [s_ * 1.0 for s_ in seeds]
        torch.autograd.backward(held["cuts"], grad_tensors=seeds, inputs=early)

    def compute():
        if flat is not None:
            flat.zero_()
        else:
            opt.zero_grad(set_to_none=True)
        with (contextlib.nullcontext() if args.no_cache else ball_query_cache()):
            ep = net(x, m, feats)
            out = head(ep) if head is not None else ep["res5_features"]
        out.square().mean().backward()
        closerlook3d_amd.join_weight_gradients()  # (nothing to do unless --weight-grads deferred)
        if world == 1:
            opt.step()

    _one_stream = {}

    def capture(fn, pool=None, warm=None, forks=None):
        """warm: what to run eagerly first (default: fn itself).  Graph B is captured right behind graph A with no eager
        step in between: it must walk the autograd graph -- and read the cut gradients -- that graph A's capture built."""
        if warm is None:
            warm = fn
        same = "other_stream" not in set(args.debug_two_graphs.split(","))  # (round 6: one stream is the default)
        if same and "s" not in _one_stream:
            _one_stream["s"] = closerlook3d_amd.step_stream(dev)
        if warm:
            side = _one_stream["s"] if same else torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    warm()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        if forks is not None:
            from closerlook3d_amd import pt_utils as _pu2
            _pu2.ASYNC_INDEX = 'auto' if forks else False
        # one graph for the whole step: declared, the engine forks what it likes.  Two graphs (--overlap): not a whole step;
        # the gradient-product forks are then an explicit choice per graph (fused.forked_gradients): on in graph A (forward +
        # late-stage backward: exact), off in graph B unless --overlap-forks b|both --unsafe asks for the known-bad layout
        with closerlook3d_amd.whole_step_capture(not args.overlap), \
                closerlook3d_amd.deferred_weight_gradients(args.weight_grads == "deferred" and not args.overlap), \
                _fu.forked_gradients(bool(forks) if (args.overlap and forks is not None) else None), \
                torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local" if world > 1 else "global",
                                 **({"stream": _one_stream["s"]} if same else {})):
            fn()
        return g

    if args.self_check:  # the cut backward against the plain one, parameter by parameter (eager, one process)
        from closerlook3d_amd.dp import FlatGradients as _FG
        flat = _FG(late + early)
        compute_late()
        compute_early()
        torch.cuda.synchronize()
        got = {n_: p.grad.detach().clone() for n_, p in list(net.named_parameters()) + ([("head." + k, v) for k, v in head.named_parameters()] if head is not None else []) if p.grad is not None}
        flat.zero_()
        with (contextlib.nullcontext() if args.no_cache else ball_query_cache()):
            ep = net(x, m, feats)
            out = head(ep) if head is not None else ep["res5_features"]
        out.square().mean().backward()
        torch.cuda.synchronize()
        worst = []
        for n_, p in list(net.named_parameters()) + ([("head." + k, v) for k, v in head.named_parameters()] if head is not None else []):
            if p.grad is None or n_ not in got:
                continue
            d = float((got[n_] - p.grad).abs().max()) / (float(p.grad.abs().max()) + 1e-30)
            worst.append((d, n_))
        worst.sort(reverse=True)
        print(json.dumps({"self_check": "cut backward vs plain backward, worst relative differences", "worst": worst[:8]}))
        return
    # the step is hundreds of short kernels: replayed as one HIP graph unless --no-graph (same kernels, same work)
    graph = update_graph = graph_b = None
    if not args.no_graph:
        try:
            if overlap:
                graph = capture(compute_late, warm=lambda: (compute_late(), compute_early()),
                                forks=args.overlap_forks in ("a", "both"))
                dbg = set(filter(None, args.debug_two_graphs.split(",")))
                if "fresh_streams" in dbg:
                    _put._INDEX_STREAMS.clear()
                _fork_count[0] = 0
                if True:
                    graph_b = capture(compute_early, pool=None if "own_pool" in dbg else graph.pool(), warm=False,
                                      forks=args.overlap_forks in ("b", "both"))
            else:
                graph = capture(compute)
            if world > 1:
                update_graph = capture(opt.step)
        except Exception as e:
            print(f"bench_backbone: HIP graph capture failed ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
            graph = update_graph = graph_b = None
            torch.cuda.synchronize()
    ar_events = []
    comm_stream = torch.cuda.Stream() if overlap else None
    from closerlook3d_amd.dp import _mean_inplace

    _dbg_sync = os.environ.get("CL3D_DP_SYNC") == "1"  # (debug: device-wide waits around the exchange)
    _dbg_noex = os.environ.get("CL3D_DP_NOEXCHANGE") == "1"  # (debug, with --repeat-check: this rank's own gradients)

    def run():
        if overlap:
            if graph is not None:
                graph.replay()
            else:
                compute_late()
            main = torch.cuda.current_stream()
            comm = main if args.comm_on_main else comm_stream
            comm.wait_stream(main)
            # the late-stage gradients leave while the early stages are differentiated: the collective is enqueued
            # behind graph A on `comm` (RCCL: the call returns at once, the stream waits for the result; gloo, the
            # one-device stand-in, blocks the host here -- correct, just not overlapped)
            with torch.cuda.stream(comm):
                _, need_div = _mean_inplace(flat.buffer[:n_late], world, None)
            if "sync_between" in set(args.debug_two_graphs.split(",")):
                torch.cuda.synchronize()
            if graph_b is not None:
                graph_b.replay()
            else:
                compute_early()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()  # from here on the exchange is exposed: what is left of the first bucket + the small second one
            _mean_inplace(flat.buffer[n_late:], world, None)
            main.wait_stream(comm)
            if need_div:
                flat.buffer.div_(world)
            e1.record()
            ar_events.append((e0, e1))
        else:
            if graph is not None:
                graph.replay()
            else:
                compute()
            if world > 1:
                if _dbg_sync:
                    torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if not _dbg_noex:
                    flat.allreduce_mean(world)
                e1.record()
                ar_events.append((e0, e1))
                if _dbg_sync:
                    torch.cuda.synchronize()
        if world > 1 and not args.repeat_check:
            if update_graph is not None:
                update_graph.replay()
            else:
                opt.step()

    if args.repeat_check and world > 1:
        seen_late, seen_early, order, per_seen = {}, {}, [], {}
        per_param = [(n_, p_) for n_, p_ in list(net.named_parameters()) + ([("head." + k, v) for k, v in head.named_parameters()] if head is not None else []) if p_.requires_grad]
        rc_trace, rc_traces = [], []
        if args.dump_forward and rank == 0:  # (eager only: which module's output first varies over the replays)
            def _rc_bits(o):
                if torch.is_tensor(o):
                    return [int(o.detach().contiguous().view(torch.int32).long().sum())] if o.dtype == torch.float32 else [int(o.long().sum())]
                if isinstance(o, dict):
                    return [b for k in sorted(o) for b in _rc_bits(o[k])]
                if isinstance(o, (tuple, list)):
                    return [b for t in o for b in _rc_bits(t)]
                return []
            for name_, mod_ in net.named_modules():
                mod_.register_forward_hook(lambda m_, i_, o_, name_=name_: rc_trace.append((name_, _rc_bits(o_))))
            if os.environ.get("CL3D_TRACE_PWMLP") == "1":  # the gather pass's operands and products, read back through HIP
                import ctypes as _ct
                from closerlook3d_amd import _lib as _cl
                _hip = _ct.CDLL("libamdhip64.so")
                _l = _cl.lib()

                def _dev_bits(ptr, nbytes):
                    torch.cuda.synchronize()
                    ptr = getattr(ptr, "value", ptr)
                    buf = np.empty(nbytes // 4, dtype=np.uint32)
                    rc_ = _hip.hipMemcpy(_ct.c_void_p(buf.ctypes.data), _ct.c_void_p(ptr), _ct.c_size_t(nbytes // 4 * 4), 2)
                    return [int(buf.astype(np.uint64).sum()), int(rc_)]
                _orig_stats = _l.cl3d_pwmlp_stats
                _first_layer = {}

                def _traced_stats(*a_):
                    q, s_, idx, ght, wr, gamma, B_, N_, M_, K_, Co_, rad, ystar, kstar, sy, partial, nparts, st_ = a_
                    tag = "stats[N=%d,Co=%d]" % (N_, Co_)
                    for nm, ptr, nb in (("q", q, B_ * M_ * 12), ("s", s_, B_ * N_ * 12), ("idx", idx, B_ * M_ * K_ * 4),
                                        ("ght", ght, B_ * N_ * 2 * Co_ * 4), ("wr", wr, Co_ * 12), ("gamma", gamma, Co_ * 4)):
                        rc_trace.append((tag + ".in." + nm, _dev_bits(ptr, nb)))
                    r_ = _orig_stats(*a_)
                    for nm, ptr, nb in (("ystar", ystar, B_ * M_ * Co_ * 4), ("kstar", kstar, B_ * M_ * Co_), ("sy", sy, B_ * M_ * Co_ * 4),
                                        ("partial", partial, nparts * Co_ * 64)):
                        rc_trace.append((tag + ".out." + nm, _dev_bits(ptr, nb)))
                    if True:  # element by element against replay 0, call by call
                        call_no = sum(1 for t_ in rc_trace if t_[0].endswith(".out.ystar"))
                        torch.cuda.synchronize()
                        cur = {}
                        for nm, ptr, n_el in (("ystar", ystar, B_ * M_ * Co_), ("sy", sy, B_ * M_ * Co_)):
                            buf = np.empty(n_el, dtype=np.float32)
                            _hip.hipMemcpy(_ct.c_void_p(buf.ctypes.data), _ct.c_void_p(ptr), _ct.c_size_t(n_el * 4), 2)
                            cur[nm] = buf.reshape(B_, M_, Co_)
                        ref = _first_layer.setdefault(call_no, cur)
                        for nm in cur:
                            d_ = np.argwhere(cur[nm].view(np.uint32) != ref[nm].view(np.uint32))
                            if len(d_):
                                bs, js, cs = np.unique(d_[:, 0]), np.unique(d_[:, 1]), np.unique(d_[:, 2])
                                print("first layer: call %d %s %s: %d elements differ from replay 0; clouds %s; %d queries %s...; channels %s; e.g. %s" % (
                                    call_no, tag, nm, len(d_), bs.tolist()[:16], len(js), js.tolist()[:24], cs.tolist(),
                                    [(tuple(int(v) for v in i_), float(ref[nm][tuple(i_)]), float(cur[nm][tuple(i_)])) for i_ in d_[:6]]),
                                    file=sys.stderr, flush=True)
                    return r_
                _l.cl3d_pwmlp_stats = _traced_stats
            import torch.autograd.function as _taf
            for cname in dir(_fu):  # every autograd Function of the engine: its outputs in call order
                cls = getattr(_fu, cname)
                if isinstance(cls, type) and issubclass(cls, _taf.Function) and cls is not _taf.Function:
                    def _wrapped(*a_, _cls=cls, _orig=cls.apply, **k_):
                        o_ = _orig(*a_, **k_)
                        shapes = [tuple(t.shape) for t in a_ if torch.is_tensor(t)][:2]
                        rc_trace.append(("%s%s" % (_cls.__name__, shapes), _rc_bits(o_)))
                        return o_
                    cls.apply = _wrapped
        for it in range(args.repeat_check):
            del rc_trace[:]
            run()
            torch.cuda.synchronize()
            rc_traces.append(list(rc_trace))
            bits = flat.buffer.view(torch.int32).long()
            kl, ke = int(bits[:n_late].sum()), int(bits[n_late:].sum())
            seen_late[kl] = seen_late.get(kl, 0) + 1
            seen_early[ke] = seen_early.get(ke, 0) + 1
            order.append((len(seen_late), len(seen_early)))
            for n_, p_ in per_param:
                per_seen.setdefault(n_, set()).add(int(p_.grad.view(torch.int32).long().sum()))
        if rank == 0:
            if args.dump_forward and rc_traces and rc_traces[0]:
                vary = [k for k in range(len(rc_traces[0])) if any(len(t) <= k or t[k] != rc_traces[0][k] for t in rc_traces[1:])]
                print(json.dumps({"forward_outputs": len(rc_traces[0]), "varying_forward_outputs": len(vary),
                                  "first_varying": [rc_traces[0][k][0] for k in vary[:12]]}), file=sys.stderr, flush=True)
            print(json.dumps({"repeat_check": args.repeat_check, "overlap": bool(overlap), "forks": args.overlap_forks,
                              "debug": args.debug_two_graphs, "comm_on_main": args.comm_on_main, "graph": graph is not None,
                              "pattern_late": max(seen_late, key=seen_late.get), "pattern_early": max(seen_early, key=seen_early.get),
                              "distinct_late": sorted(seen_late.values(), reverse=True),
                              "distinct_early": sorted(seen_early.values(), reverse=True),
                              "varying_parameters": {k: len(v) for k, v in per_seen.items() if len(v) > 1},
                              "probe": {k: int(v) for k, v in _probe.items()}}))
        dist.barrier()
        dist.destroy_process_group()
        return
    fwd_trace = []
    if args.dump_forward and rank == 0:
        def _bits(o):
            if torch.is_tensor(o):
                return [int(o.detach().contiguous().view(torch.int32).long().sum())] if o.dtype == torch.float32 else [int(o.long().sum())]
            if isinstance(o, dict):
                return [b for k in sorted(o) for b in _bits(o[k])]
            if isinstance(o, (tuple, list)):
                return [b for t in o for b in _bits(t)]
            return []
        for name_, mod_ in list(net.named_modules()) + ([("head." + k, v) for k, v in head.named_modules()] if head is not None else []):
            mod_.register_forward_hook(lambda m_, i_, o_, name_=name_: fwd_trace.append((name_, _bits(o_))))
    for it in range(args.warmup):
        run()
        if args.dump_forward and it == 0 and rank == 0:
            with open(args.dump_forward, "w") as fh:
                json.dump(fwd_trace, fh)
        if args.dump_grads and it == 0:
            torch.cuda.synchronize()
            if rank == 0:
                named = list(net.named_parameters()) + ([("head." + k, v) for k, v in head.named_parameters()] if head is not None else [])
                torch.save({k: v.grad.detach().cpu().clone() for k, v in named if v.grad is not None}, args.dump_grads)
            if world > 1:
                dist.barrier()
                dist.destroy_process_group()
            return
        if os.environ.get("CL3D_DP_DEBUG") == "1" and world > 1 and rank == 0:
            torch.cuda.synchronize()
            print("debug step: late %.9g early %.9g params %.12g" % (float(flat.buffer[:n_late].double().norm()), float(flat.buffer[n_late:].double().norm()),
                  float(torch.cat([p.detach().reshape(-1).double() for p in params]).norm())), file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    ar_events.clear()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / args.steps
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        line = {"config": args.config, "operator": kind, "n_gpus": world, "clouds_per_gpu": B, "points": N, "width": width,
                "precision": args.precision, "launch": "hip_graph" if graph is not None else "eager",
                "gemm_plans": args.gemm_plans, "weight_grads": args.weight_grads,
                "graph_queues": os.environ.get("DEBUG_HIP_FORCE_GRAPH_QUEUES", "runtime default"),
                "ms_per_step": round(dt * 1e3, 3), "input_points_per_s": round(world * B * N / dt, 1), "scaling": "weak",
                "head": "scene_seg" if head is not None else None,
                "params_M": round(sum(p.numel() for p in params) / 1e6, 2),
                "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}
        if world > 1:
            line["allreduce_exposed_ms"] = round(float(np.mean([a.elapsed_time(b) for a, b in ar_events])), 3)
            line["allreduce_bytes"] = int(flat.buffer.numel() * 4)
            line["exchange"] = ("two graphs: late-stage gradients (%d B) all-reduced while the early stages' backward "
                                "replays, remainder (%d B) after it" % (n_late * 4, (flat.buffer.numel() - n_late) * 4)
                                if overlap else "one flat all-reduce after the graph")
            line["backend"] = dist.get_backend()
            line["world_size"] = dist.get_world_size()
        line["device"] = f"cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}"
        if args.gemm_plans == "measured":
            line["gemm_plans_measured"], line["gemm_plans_changed"] = closerlook3d_amd.gemm_autotune_stats()
        if one_dev and world > 1:
            line["one_device_standin"] = True
        if args.checksums:  # same seeds, same steps: the exchange scheme must not change a bit of either
            grads = torch.cat([p.grad.reshape(-1).double() for p in params if p.grad is not None])
            line["grad_l2"] = float(grads.norm())
            line["param_l2"] = float(torch.cat([p.detach().reshape(-1).double() for p in params]).norm())
            named = list(net.named_parameters()) + ([("head." + k, v) for k, v in head.named_parameters()] if head is not None else [])
            line["grad_l2_by_param"] = [[n_, float(p_.grad.double().norm())] for n_, p_ in named if p_.grad is not None]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""bench.py -- points/sec through one LocalAggregation forward+backward (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--operator pointwisemlp|pospool|adaptive_weight|pseudo_grid]

One process per GPU (the driver launches N>1 through torch.distributed.run; backend "nccl" = RCCL).
A step = LocalAggregation fwd + bwd (+ gradient all-reduce over RCCL when N>1 + SGD update of the
operator's parameters) over one batch of B=16 synthetic clouds per GPU, N=M=4096 points, K=32
neighbours, C=64 channels (72 for PosPool, which needs C%3==0).  Clouds are resident in HBM before
the timed region.  Prints ONE JSON line on rank 0 with the contract fields plus
  roofline      top level: the timed step's longest data-moving kernel (C-ABI entry point of the fused path): algorithmic
                HBM bytes per launch / median launch duration (HIP events on the launch stream, one-stream eager run of
                the same step) against the 8 TB/s HBM peak, `traffic` = its PMC HBM bytes (profiles/rNN/
                step_counters.json), `longest_entry` beside it when a VALU-bound entry (the ball query) is longer;
                `boundary`: the reference-visible `_ext` MATERIALISING path (`path` says so) -- MaskedQueryAndGroup and
                its backward as the reference's own Python calls them -- per kernel (median / min / max over >= 12
                bursts) and as a whole at SURVEY 8(d)'s 17,696 B/point (north_star's >= 50 % target); these kernels are
                NOT the ones the timed step runs (impl=auto takes the fused path);
                `step`: the timed step's OWN kernels, one row per C-ABI entry point (calls, microseconds, algorithmic
                HBM bytes, PMC HBM / L2 bytes when committed, bound, fractions);
                `achieved_step` = value x 17,696 B / 8 TB/s (what the headline corresponds to);
                `contraction`: the hand-written MFMA per-point GEMMs of the PointWiseMLP (flops, microseconds,
                fraction of the 157.3 TFLOP/s f32-input / 2.5 PFLOP/s bf16 dense MFMA peaks), with the vendor
                library's time for the same three products beside it
  cpu_baseline  the CPU oracle (C restatement of the native ops, OpenMP over batch x query, + torch CPU operators)
                on a bounded sample of the same workload, rank 0, N=1 only: `value`/`cores` = every host core,
                `one_thread` = the same on a single thread (SURVEY 8(d) asks for both).
"""
import argparse
import glob
import json
import os
import sys
import time

# before the HIP runtime starts, whoever launched this rank (the driver's own `torch.distributed.run ... bench.py --gpus N`
# included): dmabuf IPC, which RCCL needs on this driver (closerlook3d_amd/dp.py prepare_environment does the same)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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
    if os.environ.get("CL3D_BENCH_SORTED") == "1":
        # experiment: points stored in cell order (cells of 0.14, z-y-x major) -- what processing tiles in the ball
        # query's cell order would buy the gather passes, without touching them
        cell = np.floor(xyz / 0.14).astype(np.int64)
        key = (cell[..., 2] * 64 + cell[..., 1]) * 64 + cell[..., 0]
        order = np.argsort(key, axis=1, kind="stable")
        xyz = np.take_along_axis(xyz, order[..., None], axis=1)
    feats = rng.standard_normal((B, C, N)).astype(np.float32)
    return xyz, mask, feats


def event_time_stats(fn, bursts=12, warmup=3, burst=8):
    """Duration of one fn() in ms over `bursts` bursts of `burst` back-to-back launches (HIP events on the current
    (= launch) stream around each burst, so the device stays busy and host launch latency is not counted):
    {median, min, max, bursts} of the per-burst averages.  The count does not depend on --steps (two
    bursts made the line's fraction irreproducible)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    start = [torch.cuda.Event(enable_timing=True) for _ in range(bursts)]
    stop = [torch.cuda.Event(enable_timing=True) for _ in range(bursts)]
    for i in range(bursts):
        start[i].record()
        for _ in range(burst):
            fn()
        stop[i].record()
    torch.cuda.synchronize()
    per = sorted(s.elapsed_time(e) / burst for s, e in zip(start, stop))
    return {"median": float(np.median(per)), "min": per[0], "max": per[-1], "bursts": bursts, "launches_per_burst": burst}


def kernel_rooflines(xyz, mask, feats, radius, K, bursts):
    """Per-kernel achieved bandwidth of the materialising ball_query+group path; every figure is the
    MEDIAN over `bursts` bursts, with the spread beside it."""
    from closerlook3d_amd import _ext
    B, N, _ = xyz.shape
    C = feats.shape[1]
    M = N
    idx, _ = _ext.masked_ordered_ball_query(xyz, xyz, mask, mask, radius, K)
    grad_out = torch.randn(B, C, M, K, device=xyz.device)
    MK = M * K
    specs = {
        # name: (callable, algorithmic bytes per launch)
        "ball_query": (lambda: _ext.masked_ordered_ball_query(xyz, xyz, mask, mask, radius, K),
                       B * (12 * M + 12 * N + 4 * M + 4 * N + 8 * MK)),
        "group_fwd_lds_kernel": (lambda: _ext.group_points(feats, idx), B * (4 * C * MK + 4 * C * N + 4 * MK)),
        "group_rel_kernel+group_fwd_lds_kernel": (lambda: _ext.group_xyz_features(xyz, xyz, feats, idx, radius, True),
                                                  B * (12 * M + 12 * N + 4 * C * N + 4 * MK + 12 * MK + 4 * C * MK)),
        "group_bwd_lds_kernel": (lambda: _ext.group_points_grad(grad_out, idx, N), B * (4 * C * MK + 4 * MK + 4 * C * N)),
    }
    out = {}
    for name, (fn, nbytes) in specs.items():
        st = event_time_stats(fn, bursts)
        ms = st["median"]
        out[name] = {"ms": round(ms, 5), "ms_min": round(st["min"], 5), "ms_max": round(st["max"], 5),
                     "bursts": st["bursts"], "bytes": int(nbytes), "achieved_GBps": round(nbytes / ms / 1e6, 1),
                     "frac": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4)}
    # the whole reference-visible boundary: fwd (query + group) and bwd (scatter), 17,696 B/point at the metric shape
    parts = ("ball_query", "group_rel_kernel+group_fwd_lds_kernel", "group_bwd_lds_kernel")
    fwd = 12 * M + 12 * N + 4 * M + 4 * N + 4 * C * N + 4 * MK + 4 * MK + 12 * MK + 4 * C * MK
    bwd = 4 * C * MK + 4 * MK + 4 * C * N

    def frac_at(key):
        total = sum(out[p][key] for p in parts)
        return total, B * (fwd + bwd) / (total * 1e-3) / HBM_PEAK

    total_ms, frac = frac_at("ms")
    _, frac_slowest = frac_at("ms_max")   # every part at its slowest burst: the floor of the fraction
    _, frac_fastest = frac_at("ms_min")
    boundary = {"ms": round(total_ms, 5), "bytes_per_point": (fwd + bwd) / M,
                "points_per_s": round(B * M / (total_ms * 1e-3), 1),
                "achieved_GBps": round(B * (fwd + bwd) / total_ms / 1e6, 1),
                "frac": round(frac, 4), "frac_min": round(frac_slowest, 4), "frac_max": round(frac_fastest, 4),
                "frac_of_measured_copy_peak": round(frac * HBM_PEAK / HBM_MEASURED, 4),
                "definition": "sum of the medians of ball_query + (group_rel + group_fwd) + group_bwd; frac_min takes "
                              "every part at its slowest burst"}
    return out, boundary


L2_PEAK = 34.5e12  # B/s aggregate L2 bandwidth (MI355X_MICROARCH.md, "L2 (per XCD)")

# C-ABI entry point -> kernels it launches (names as rocprofv3 prints them), for merging PMC counters
ENTRY_KERNELS = {
    "cl3d_masked_ordered_ball_query": ["bq_tile_kernel", "bq_prep_kernel", "bq_query_kernel", "ball_query_kernel"],
    "cl3d_build_inverse_index": ["csr_count_fill_kernel", "csr_rows_kernel", "csr_scan_kernel"],
    # (f32: the LDS-free kernel; bf16 and other shapes: mfma_gemm_kernel, see entry_kernels)
    "cl3d_pwmlp_point_gemm_fwd": ["pwmlp_weights_kernel", "pwmlp_rows_nolds_kernel"],
    "cl3d_pwmlp_point_gemm_bwd_data": ["mfma_gemm_kernel"],
    "cl3d_pwmlp_point_gemm_bwd_weight": ["mfma_gemm_kernel", "gemm_reduce_kernel"],
    "cl3d_pwmlp_point_gemm_bwd": ["pwmlp_point_grads_kernel", "pwmlp_dw_reduce_kernel"],
    "cl3d_pwmlp_stats": ["pwmlp_query_kernel<0"],
    "cl3d_pwmlp_finalize_stats": ["pwmlp_finalize_kernel<0"],
    "cl3d_pwmlp_apply": ["pwmlp_rows64_kernel<0", "pwmlp_rows_kernel<0"],   # whole-tile form first (the metric shape)
    "cl3d_pwmlp_bwd_rows": ["pwmlp_rows64_kernel<1", "pwmlp_rows_kernel<1"],
    "cl3d_pwmlp_bwd_hits": ["pwmlp_hit_kernel"],
    "cl3d_pwmlp_bn_backward_coeffs": ["pwmlp_finalize_kernel<1"],
    "cl3d_pwmlp_bwd_hits_coeffs": ["pwmlp_hit_coeffs_kernel"],  # both of the above in one launch (the training step)
    "cl3d_pwmlp_bwd_support": ["pwmlp_support_kernel"],
    "cl3d_fused_reduce_fwd": ["fused_reduce_fwd_kernel"],
    "cl3d_fused_reduce_bwd": ["fused_reduce_bwd_kernel", "pg_dkw_kernel"],
    "cl3d_transpose": ["transpose_kernel", "transpose4_kernel"],
    "cl3d_bn_relu_stats": ["bn_stats_kernel<0", "bn_finalize_kernel<0"],
    "cl3d_bn_relu_apply": ["bn_apply_kernel<0"],
    "cl3d_bn_relu_bwd": ["bn_stats_kernel<1", "bn_finalize_kernel<1", "bn_apply_kernel<1"],
    "cl3d_fused_param_reduce": ["param_reduce_kernel"],
}


def entry_kernels(entry, counters):
    """Kernel-name prefixes of one C-ABI entry point, given the kernels a PMC session actually saw."""
    prefs = list(ENTRY_KERNELS.get(entry, []))
    if entry == "cl3d_pwmlp_point_gemm_fwd" and not any("pwmlp_rows_nolds_kernel" in k for k in counters):
        prefs.append("mfma_gemm_kernel")
    return prefs


def step_model_bytes(B, N, M, K, C, kind="pointwisemlp"):
    """Per entry point: (algorithmic HBM bytes = every distinct input byte read once + every output byte written
    once, modelled L2 gather bytes = rows fetched through the cache hierarchy by the gather passes, bound)."""
    MK, Co = M * K, C
    f = 4
    xyzm = 12 * (M + N) + 4 * (M + N)
    rows_q = f * M * Co
    pg = 1 if kind == "pseudo_grid" else 0  # PseudoGrid keeps 32 B of (kernel point, influence) pairs per slot
    t = {
        "cl3d_masked_ordered_ball_query": (B * (xyzm + 8 * MK), 0, "valu+latency"),
        "cl3d_build_inverse_index": (B * (8 * MK + 4 * N), 0, "latency"),
        "cl3d_pwmlp_point_gemm_fwd": (B * f * (C * N + N * 2 * Co), 0, "mfma+hbm"),
        "cl3d_pwmlp_point_gemm_bwd_data": (B * f * (C * N + N * 2 * Co), 0, "mfma+hbm"),
        "cl3d_pwmlp_point_gemm_bwd_weight": (B * f * (C * N + N * 2 * Co), 0, "mfma+hbm"),
        # both gradients from one pass: d ght and the features read once, d features written once
        "cl3d_pwmlp_point_gemm_bwd": (B * f * (2 * C * N + N * 2 * Co), 0, "mfma+hbm"),
        # TRAIN gather pass: one G row (Co floats) per slot + the centre's H row per query
        "cl3d_pwmlp_stats": (B * (f * N * 2 * Co + 4 * MK + xyzm + 2 * rows_q + M * Co), B * MK * f * Co, "l2-gather+latency"),
        "cl3d_pwmlp_apply": (B * 2 * rows_q, 0, "hbm"),
        "cl3d_pwmlp_bwd_rows": (B * (4 * rows_q + M * Co + 4 * MK + xyzm), 0, "hbm"),
        "cl3d_pwmlp_bwd_hits": (B * (2 * rows_q + f * Co * N), 0, "lds-atomics"),
        "cl3d_pwmlp_bwd_hits_coeffs": (B * (2 * rows_q + f * Co * N), 0, "lds-atomics"),
        # support-major pass: one H row per slot through the CSR (slot ids, row bounds, the per-query table of bwd_rows)
        "cl3d_pwmlp_bwd_support": (B * (f * N * 2 * Co * 2 + f * Co * N + 2 * rows_q + 4 * MK + 4 * N + 16 * M + 12 * N), B * MK * f * Co, "l2-gather+latency"),
        # PosPool / AdaptiveWeight / PseudoGrid: one feature row per slot each way
        "cl3d_fused_reduce_fwd": (B * (f * C * N + xyzm + 8 * MK + f * C * M + 16 * MK + pg * 32 * MK), B * MK * f * C, "l2-gather+latency"),
        "cl3d_fused_reduce_bwd": (B * (3 * f * C * N + f * C * M + 16 * MK + 8 * MK + pg * 64 * MK), B * MK * f * C, "l2-gather+latency"),
        # the parameter gradients of AdaptiveWeight / PseudoGrid: block partials [nparts, C, 4 | 16] summed in one launch
        "cl3d_fused_param_reduce": (0, 0, "small"),
        "cl3d_transpose": (2 * B * f * C * N, 0, "hbm"),
        "cl3d_bn_relu_stats": (B * f * C * N, 0, "hbm"),
        "cl3d_bn_relu_apply": (2 * B * f * C * N, 0, "hbm"),
        "cl3d_bn_relu_bwd": (4 * B * f * C * N, 0, "hbm"),
    }
    return t


def step_counters(kind="pointwisemlp"):
    """Per-kernel PMC sums of the bench step committed under profiles/rNN/step_counters.json (step_counters_<operator>.json
    for the other three operators; scripts/step_counters.py: separate rocprofv3 --pmc passes, gfx950 FETCH_SIZE
    correction), newest round; {} when absent."""
    name = "step_counters.json" if kind == "pointwisemlp" else f"step_counters_{kind}.json"
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)))
    if not paths:
        return {}, None
    try:
        return json.load(open(paths[-1]))["kernels"], os.path.relpath(paths[-1], ROOT)
    except Exception:
        return {}, None


def step_table(compute, B, N, M, K, C, reps, kind="pointwisemlp"):
    """The timed step's own kernels: GPU microseconds per C-ABI entry point from HIP events on the launch stream, in
    an eager run of the SAME compute() with the index streams folded onto the main stream (so durations are not
    stretched by overlap); median / min / max over `reps` runs.  The graph-replayed step overlaps some of these, so
    the rows sum to more than ms_per_step.  L2 figures come from the committed PMC passes (TCC_HIT + TCC_MISS
    requests x 128 B), not from a model."""
    from closerlook3d_amd import _lib, pt_utils
    saved = pt_utils.ASYNC_INDEX
    pt_utils.ASYNC_INDEX = False
    try:
        for _ in range(2):
            compute()
        with _lib.trace() as tr:
            for _ in range(reps):
                compute()
        per_run = tr.per_run(reps)
    finally:
        pt_utils.ASYNC_INDEX = saved
    model = step_model_bytes(B, N, M, K, C, kind)
    counters, counters_src = step_counters(kind)
    rows = []
    for name, (calls, runs) in per_run.items():
        us_step = float(np.median(runs))
        alg, l2, bound = model.get(name, (0, 0, "small"))
        alg = int(alg * calls)
        row = {"entry": name, "calls": calls, "us": round(us_step, 2), "us_min": round(min(runs), 2),
               "us_max": round(max(runs), 2), "algorithmic_bytes": alg,
               "hbm_frac": round(alg / (us_step * 1e-6) / HBM_PEAK, 4) if us_step > 0 else None, "bound": bound,
               "kernels": entry_kernels(name, counters)}
        hb = lb = 0.0
        found = False
        for kname, rec in counters.items():
            if any(kname.replace("cl3d::", "").startswith(pref) for pref in entry_kernels(name, counters)):
                hb += rec.get("hbm_bytes", 0.0)
                lb += rec.get("l2_bytes", 0.0)
                found = True
        if found:
            row["hbm_bytes_pmc"], row["l2_bytes_pmc"] = int(hb), int(lb)
            row["l2_frac"] = round(lb / (us_step * 1e-6) / L2_PEAK, 4) if us_step > 0 else None
        rows.append(row)
    rows.sort(key=lambda r: -r["us"])
    return {"source": "HIP events around every C-ABI call on the launch stream, eager one-stream run of the timed step, "
                      "median of %d runs" % reps,
            "pmc_source": counters_src, "sum_us": round(sum(r["us"] for r in rows), 1),
            "dominant": rows[0]["entry"] if rows else None, "kernels": rows}


def contraction_block(B, C, N, Co, precision, reps=30):
    """roofline.contraction: the hand-written MFMA per-point GEMMs at this step's shape (scripts/bench_point_gemm.py)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench_point_gemm
    m = bench_point_gemm.measure(B, C, N, Co, reps=reps)
    mine = m["mfma_" + precision]
    # what the timed step runs: the forward product and ONE call that forms both gradients (one kernel over d ght where
    # pwmlp_point_grads_kernel covers the shape -- f32 at the metric shape -- else the two products in a row)
    us = mine["fwd_us"] + mine["bwd_both_us"]
    peak = mine["peak_tflops"]
    out = {"kernel": "cl3d::pwmlp_rows_nolds_kernel (forward) + cl3d::pwmlp_point_grads_kernel (both gradients) + "
                     "cl3d::pwmlp_dw_reduce_kernel; cl3d::mfma_gemm_kernel where they do not cover the shape (csrc/mfma_gemm.hip)",
           "precision": precision,
           "instruction": "v_mfma_f32_32x32x2_f32" if precision == "f32" else "v_mfma_f32_32x32x16_bf16",
           "flops": 3 * m["flops_per_gemm"], "us": round(us, 2),
           "separate_products_us": round(mine["fwd_us"] + mine["bwd_data_us"] + mine["bwd_weight_us"], 2),
           "fwd_us": round(mine["fwd_us"], 2), "bwd_data_us": round(mine["bwd_data_us"], 2),
           "bwd_weight_us": round(mine["bwd_weight_us"], 2),
           "bwd_both_us": round(mine["bwd_both_us"], 2), "bwd_both_one_kernel": mine["bwd_both_one_kernel"],
           "achieved_TFLOPs": round(3 * m["flops_per_gemm"] / (us * 1e-6) / 1e12, 2), "peak_TFLOPs": peak,
           "frac": round(3 * m["flops_per_gemm"] / (us * 1e-6) / 1e12 / peak, 4),
           # bytes the step's two calls must move: features + ght (forward), d ght + features + d features (gradients)
           "hbm_bytes": int(4 * B * N * (2 * C + 2 * 2 * Co + C)),
           "hbm_frac": round(4 * B * N * (2 * C + 2 * 2 * Co + C) / (us * 1e-6) / HBM_PEAK, 4),
           "note": "2*B*N*C*2Co flops per product, three products (ght, d features, d weight); each time includes the "
                   "call's small side launch (weight split / partial reduce); `separate_products_us` = the three products as "
                   "three calls (round 4's form)"}
    other = "bf16" if precision == "f32" else "f32"
    o = m["mfma_" + other]
    out["other_precision"] = {"precision": other, "us": round(o["fwd_us"] + o["bwd_data_us"] + o["bwd_weight_us"], 2),
                              "frac": round(o["frac_of_mfma_peak"], 4)}
    if "library_f32" in m:
        lb = m["library_f32"]
        out["library_f32_us"] = {"fwd": round(lb["fwd_us"], 2), "bwd_data": round(lb["bwd_data_us"], 2),
                                 "bwd_weight": round(lb["bwd_weight_us"], 2),
                                 "total": round(lb["fwd_us"] + lb["bwd_data_us"] + lb["bwd_weight_us"], 2)}
    return out


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC passes (scripts/pmc_kernels.py; rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 FETCH_SIZE correction applied), or None.  "ball_query" sums the
    kernels one call of the op launches."""
    prefixes = ("bq_", "ball_query_kernel") if kernel == "ball_query" else (kernel,)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")), reverse=True):
        try:
            data = json.load(open(path))["kernels"]
        except Exception:
            continue
        hits = [rec["hbm_bytes"] for name, rec in data.items() if name.replace("cl3d::", "").startswith(prefixes)]
        if hits:
            return float(sum(hits))
    return None


def cpu_baseline_child(kind, N, K, C, radius, clouds, iters, threads):
    """One leg of the CPU baseline in a process of its own (so that its OpenMP binding touches nothing else): the
    oracle (port of the reference semantics) timed on the host, LA fwd+bwd over `clouds` clouds with `threads`
    threads -- OpenMP over batch x query in the C restatement of the native ops + torch's intra-op threads."""
    from oracle import native as on  # noqa: F401  (test/bench infrastructure only)
    from oracle import operators as oo
    torch.set_num_threads(threads)
    on.set_threads(threads)
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
    print(json.dumps({"seconds": float(np.median(ts)), "threads": threads, "clouds": clouds}), flush=True)


def cpu_baseline(kind, N, K, C, radius, clouds, iters, threads_sweep=(8, 16, 32, 64)):
    """the reference's CPU-side equivalent timed on the host cores of the same box.  Every leg runs in a
    child process with OMP_PLACES=cores OMP_PROC_BIND=close (threads on physical cores, close together).  all_cores =
    the best of a small sweep over the thread count -- the operator's tensors stop scaling long before every core is
    busy (round 2 timed 128 + 128 threads on 4 clouds and got LESS than one thread); one_thread = the literal
    one-block-per-cloud structure of the reference kernels, on a quarter of the sample."""
    import subprocess

    def leg(n_clouds, n_iters, threads):
        env = dict(os.environ, OMP_PLACES="cores", OMP_PROC_BIND="close", OMP_NUM_THREADS=str(threads),
                   HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child",
               json.dumps([kind, N, K, C, radius, n_clouds, n_iters, threads])]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            raise RuntimeError("cpu baseline leg failed: " + r.stderr[-500:])
        return json.loads(r.stdout.strip().splitlines()[-1])["seconds"]

    phys = physical_cores()
    sweep = {nt: leg(clouds, iters, nt) for nt in ([t for t in threads_sweep if t <= phys] or [phys])}
    best_t = min(sweep, key=sweep.get)
    med_all = sweep[best_t]
    quarter = max(1, clouds // 4)
    med_one = leg(quarter, 1, 1)
    try:
        cpu_model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        cpu_model = "unknown"
    return {"value": round(clouds * N / med_all, 1), "unit": "points/s", "cores": best_t,
            "kind": "port",
            "sample": f"{clouds} clouds x {iters} timed fwd+bwd of the same operator/shape (N={N},K={K},C={C}); "
                      f"C oracle for the native ops (OpenMP over batch x query) + torch CPU ops, {best_t} threads each "
                      f"(best of the sweep), OMP_PLACES=cores OMP_PROC_BIND=close; median {med_all * 1e3:.1f} ms; host: "
                      f"{cpu_model}, {phys} physical / {os.cpu_count()} logical cores",
            "all_cores": {"value": round(clouds * N / med_all, 1), "ms": round(med_all * 1e3, 1),
                          "omp_threads": best_t, "torch_threads": best_t,
                          "sweep_ms": {str(k): round(v * 1e3, 1) for k, v in sweep.items()}},
            "one_thread": {"value": round(quarter * N / med_one, 1), "ms": round(med_one * 1e3, 1), "clouds": quarter,
                           "omp_threads": 1, "torch_threads": 1}}


def physical_cores():
    """Physical cores this process may run on (distinct (package, core) pairs of the allowed CPUs)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1
    seen = set()
    for c in allowed:
        try:
            pkg = open(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id").read().strip()
            core = open(f"/sys/devices/system/cpu/cpu{c}/topology/core_id").read().strip()
            seen.add((pkg, core))
        except OSError:
            seen.add(("?", c))
    return max(len(seen), len(allowed) // 2)  # virtual machines report one (package, core) pair for every CPU


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


# "same": warm-up and capture on closerlook3d_amd.step_stream() (round 6); "separate": torch's usual two streams (the A/B arm,
# --capture-stream separate)
CAPTURE_STREAM = "same"

BACKBONE_OF = {"pointwisemlp": ("modelnet_pointwisemlp", "bf16"), "pseudo_grid": ("s3dis_pseudogrid", "f32"),
               "adaptive_weight": ("partnet_adaptive", "f32"), "pospool": ("s3dis_pospool_deep", "f32")}


def backbone_step(kind, world, rank, dev, steps=10, warmup=3):
    """SURVEY 8(e)'s second figure, beside the operator-only headline: one training step of the BASELINE.json BACKBONE
    this operator belongs to (pointwisemlp -> config 2: 16 x 4096 points, width 144, bf16 contractions) -- forward,
    backward, the gradient mean of EVERY parameter over all ranks (one flat RCCL all-reduce of 74-106 MB, the exchange
    the reference's DistributedDataParallel makes, function/train_modelnet_dist.py:206,280), SGD update.  The compute is
    one HIP graph, the all-reduce sits between it and the update graph.  Measured after the headline's timed region;
    weak scaling (every rank its own clouds).  Returns the dict that goes into the JSON line."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from bench_backbone import CONFIGS
    import closerlook3d_amd
    from closerlook3d_amd.backbones import ResNet
    from closerlook3d_amd.dp import FlatGradients
    from closerlook3d_amd.pt_utils import ball_query_cache
    name, precision = BACKBONE_OF[kind]
    bkind, B, N, radius, dl, nsamples, npoints, width = CONFIGS[name]
    import closerlook3d_amd
    closerlook3d_amd.gemm_autotune(True)  # tile / K-slice plans of the dense products timed during the warm-up steps
    torch.manual_seed(0)
    cfg = make_config(bkind, "auto")
    cfg["cl3d_precision"] = precision
    if bkind == "pospool" and "deep" in name:
        cfg.pospool.position_embedding = "sin_cos"
    net = ResNet(cfg, 3, radius, dl, nsamples, npoints, width=width, depth=2, bottleneck_ratio=2).to(dev).train(True)
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3)
    xyz, mask, _ = synth_batch(B, N, 3, 7 + rank)
    scale = 1.0 if N <= 16384 else 4.0
    x = torch.from_numpy((xyz * scale).astype(np.float32)).to(dev)
    m = torch.from_numpy(mask).to(dev)
    feats = x.transpose(1, 2).contiguous()
    flat = FlatGradients(params) if world > 1 else None

    def compute():
        if flat is not None:
            flat.zero_()
        else:
            opt.zero_grad(set_to_none=True)
        with ball_query_cache():
            out = net(x, m, feats)["res5_features"]
        out.square().mean().backward()
        closerlook3d_amd.join_weight_gradients()  # (deferred inside the capture below: one join, in front of the exchange)
        if world == 1:
            opt.step()

    def capture(fn):
        # warm-up and capture on ONE stream (closerlook3d_amd.step_stream: autograd's AccumulateGrad nodes then run on the
        # capture stream instead of forming a third branch of the captured backward pass)
        side = closerlook3d_amd.step_stream(dev) if CAPTURE_STREAM == "same" else torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with closerlook3d_amd.whole_step_capture(), closerlook3d_amd.deferred_weight_gradients(), \
                torch.cuda.graph(g, capture_error_mode="thread_local" if world > 1 else "global",
                                 **({"stream": side} if CAPTURE_STREAM == "same" else {})):
            fn()
        return g

    graph = update_graph = None
    failure = None
    try:
        try:
            graph = capture(compute)
            if world > 1:
                update_graph = capture(opt.step)
        except torch.OutOfMemoryError:
            raise
        except Exception as e:
            print(f"bench: backbone step: HIP graph capture failed ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
            graph = update_graph = None
            torch.cuda.synchronize()
            compute()  # (eager launches must at least run once on this rank before the ranks meet in a collective)
            torch.cuda.synchronize()
    except Exception as e:  # out of memory, a kernel's error code: this rank cannot run the step
        failure = f"{type(e).__name__}: {e}"
    if world > 1:
        # one rank failing before the first collective must not leave the others in the all-reduce below.  The
        # ranks agree on a flag here (the only collective a failed rank still joins); on failure every rank returns the
        # error record and the headline, already measured, is printed with it.
        ok = torch.tensor([0 if failure else 1], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            return {"config": name, "error": failure or "another rank could not set the step up (its stderr has the reason)"}
    elif failure:
        return {"config": name, "error": failure}
    ar = []

    def step():
        graph.replay() if graph is not None else compute()
        if world > 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            flat.allreduce_mean(world)
            e1.record()
            ar.append((e0, e1))
            update_graph.replay() if update_graph is not None else opt.step()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    del ar[:]
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / steps
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    out = {"config": name, "what": "5-stage residual backbone step: forward + backward + gradient mean over all ranks + SGD",
           "precision": precision, "clouds_per_gpu": B, "points": N, "width": width, "steps": steps,
           "launch": "hip_graph" if graph is not None else "eager", "ms_per_step": round(dt * 1e3, 3),
           "input_points_per_s": round(world * B * N / dt, 1), "scaling": "weak",
           "params_M": round(sum(p.numel() for p in params) / 1e6, 2),
           "gemm_plans": "measured during the warm-up steps (closerlook3d_amd.gemm_autotune): %d products, %d off the model's plan"
                         % closerlook3d_amd.gemm_autotune_stats(),
           "weight_grads": "deferred" if graph is not None else "joined",
           "graph_queues": os.environ.get("DEBUG_HIP_FORCE_GRAPH_QUEUES", "runtime default")}
    if world > 1:
        out["allreduce_bytes"] = int(flat.buffer.numel() * 4)
        out["allreduce_ms"] = round(float(np.mean([a.elapsed_time(b) for a, b in ar])), 3)
        out["exchange"] = "one flat all-reduce (mean) of every parameter gradient between the step graph and the update graph"
    return out


def backbone_step_child(kind, steps=10):
    """The N = 1 form of backbone_step(): the same step (scripts/bench_backbone.py builds the same network, batch, graph and
    timing loop) in a process of its own, started after the headline's timed region, so that it can run under the graph
    layout that is fastest FOR IT without touching the headline's: DEBUG_HIP_FORCE_GRAPH_QUEUES=3 -- the number of hardware
    queues the HIP runtime lays a captured graph out on, read once when the runtime starts -- takes the config-2 step from
    5.63 to 5.40 ms and leaves the 16-kernel operator step where it is or 0.3 % slower (profiles/r06/session33_summary.txt);
    and with the contractions' weight gradients joined once in front of the optimizer instead of layer by layer
    (closerlook3d_amd.deferred_weight_gradients).  An explicit DEBUG_HIP_FORCE_GRAPH_QUEUES in the environment is kept."""
    import subprocess
    name, precision = BACKBONE_OF[kind]
    env = dict(os.environ)
    env.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "3")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "bench_backbone.py"), "--config", name, "--precision", precision,
           "--steps", str(steps), "--warmup", "3", "--weight-grads", "deferred"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError("bench_backbone.py exited with %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:]))
    d = json.loads(lines[-1])
    out = {"config": name, "what": "5-stage residual backbone step: forward + backward + gradient mean over all ranks + SGD",
           "precision": precision, "clouds_per_gpu": d["clouds_per_gpu"], "points": d["points"], "width": d["width"],
           "steps": steps, "launch": d["launch"], "ms_per_step": d["ms_per_step"], "input_points_per_s": d["input_points_per_s"],
           "scaling": "weak", "params_M": d["params_M"],
           "gemm_plans": "measured during the warm-up steps (closerlook3d_amd.gemm_autotune): %d products, %d off the model's plan"
                         % (d.get("gemm_plans_measured", 0), d.get("gemm_plans_changed", 0)),
           "weight_grads": d.get("weight_grads"), "graph_queues": d.get("graph_queues"),
           "process": "child (scripts/bench_backbone.py) started after the headline's timed region, DEBUG_HIP_FORCE_GRAPH_QUEUES="
                      + env["DEBUG_HIP_FORCE_GRAPH_QUEUES"]}
    return out


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-baseline-child":
        return cpu_baseline_child(*json.loads(sys.argv[2]))
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
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"],
                    help="arithmetic of the PointWiseMLP's dense contraction (bf16 inputs to the MFMA, f32 accumulation)")
    ap.add_argument("--no-step-table", action="store_true")
    ap.add_argument("--precondition", type=int, default=300,
                    help="untimed steps before the warm-up (clock ramp of an idle device; not part of --warmup / --steps)")
    ap.add_argument("--bursts", type=int, default=12, help="bursts of 8 launches per boundary kernel (median / min / max reported)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a HIP graph")
    ap.add_argument("--optimizer", default="torch", choices=["torch", "flat"],
                    help="torch: torch.optim.SGD, the reference's optimizer (default); flat: the engine's one-launch update over flat "
                         "buffers (closerlook3d_amd.optim.FlatSGD) -- same arithmetic, no Python per parameter: for eager launches")
    ap.add_argument("--capture-stream", default="same", choices=["same", "separate"],
                    help="same (default): warm-up and capture on one stream; separate: a warm-up stream of its own (A/B)")
    ap.add_argument("--backbone", default="auto", choices=["auto", "on", "off"],
                    help="also time the BASELINE backbone step incl. the gradient all-reduce (SURVEY 8(e)): 'backbone_step' in "
                         "the JSON line; auto = on unless --no-kernel-roofline asks for a bare run")
    ap.add_argument("--backbone-steps", type=int, default=10)
    args = ap.parse_args()
    global CAPTURE_STREAM
    CAPTURE_STREAM = args.capture_stream

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # typed into a bare shell: become the N-rank launch of this very command (one process per GPU over RCCL; on a
        # box with fewer devices every rank shares device 0 over gloo and the line says so)
        from closerlook3d_amd.dp import self_launch
        self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU (python bench.py --gpus N "
                         f"does it itself when WORLD_SIZE is unset)")
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

    import closerlook3d_amd
    from closerlook3d_amd.dp import FlatGradients, device_identity, rank_census
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation

    # who sits where: raises under RCCL when two ranks share a device (that is not a scaling run)
    census = rank_census(device_identity(dev)) if world > 1 else [{"rank": 0, "device": device_identity(dev)}]

    kind = args.operator
    B, N, K = args.batch, args.points, args.nsample
    C = args.channels or (72 if kind == "pospool" else 64)
    radius = float((1.5 * K * 3 / (4 * np.pi * N)) ** (1 / 3))  # mean in-radius count ~1.5K
    if kind == "pseudo_grid":
        radius = float((1.5 * K * 3 / (4 * np.pi * N)) ** (1 / 3))
    torch.manual_seed(0)  # same parameters on every rank
    cfg = make_config(kind, args.impl)
    cfg["cl3d_precision"] = args.precision
    module = LocalAggregation(C, C, radius, K, cfg).to(dev).train(True)
    params = [p for p in module.parameters() if p.requires_grad]
    if args.optimizer == "flat" and params:
        # the engine's one-launch update over flat buffers (closerlook3d_amd/optim.py): same arithmetic, one C-ABI call
        # instead of torch.optim.SGD's Python -- what an EAGERLY launched step gains (the replayed step has no host side)
        from closerlook3d_amd.optim import FlatSGD
        opt = FlatSGD(params, lr=1e-3)
    else:
        opt = torch.optim.SGD(params, lr=1e-3) if params else None
    xyz, mask, feats = (torch.from_numpy(a).to(dev) for a in synth_batch(B, N, C, 1000 + rank))
    feats.requires_grad_(True)
    probe = torch.randn(B, C, N, device=dev)

    # N > 1: the parameter gradients live in one flat buffer (zeroed inside the captured step, exchanged with a
    # single in-place RCCL all-reduce) and the update is a second, tiny graph -- three host calls per step
    flat = FlatGradients(params) if (world > 1 and params and args.optimizer != "flat") else None
    flat_opt_grads = opt.flat_grads[0] if (world > 1 and params and args.optimizer == "flat") else None

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
        side = closerlook3d_amd.step_stream(dev) if CAPTURE_STREAM == "same" else torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # N > 1: RCCL's watchdog thread polls events while this thread captures; thread-local capture mode keeps
        # another thread's event query from invalidating the capture (this thread makes no unsafe call itself)
        # (forward and backward of the step in ONE capture: the operator may leave its forked geometry work to be
        # joined by its backward, closerlook3d_amd.whole_step_capture)
        with closerlook3d_amd.whole_step_capture(), \
                torch.cuda.graph(g, capture_error_mode="thread_local" if world > 1 else "global",
                                 **({"stream": side} if CAPTURE_STREAM == "same" else {})):
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
            elif flat_opt_grads is not None:
                dist.all_reduce(flat_opt_grads)
                flat_opt_grads.div_(world)
            if update_graph is not None:
                update_graph.replay()
            elif opt is not None:
                opt.step()

    # Device preconditioning, untimed and outside the W warm-up steps: a step is 0.37 ms, so W = 5 steps are 2 ms --
    # not enough for the clocks of a box that sat idle (the round-2 driver run, --steps 20 --warmup 5, read 0.396 ms
    # where 100-step runs on the same build read 0.383).  ~0.1 s of replays first, then the warm-up the caller asked for.
    for _ in range(args.precondition):
        step()
    torch.cuda.synchronize()
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

    want_backbone = args.backbone == "on" or (args.backbone == "auto" and not args.no_kernel_roofline)
    bb = None
    if want_backbone:  # (a collective when N > 1: every rank runs it, after the headline's timed region)
        try:
            bb = (backbone_step_child(kind, steps=args.backbone_steps) if world == 1
                  else backbone_step(kind, world, rank, dev, steps=args.backbone_steps))
        except Exception as e:  # the headline stands on its own; say what happened
            # (set-up failures of any rank come back as an agreed {"error": ...}; what still raises here for N > 1 happened
            # between collectives, where the ranks cannot be re-joined -- rank 0 prints its headline before giving up)
            bb = {"error": f"{type(e).__name__}: {e}"}
            if world > 1:
                bb["fatal"] = True
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * N / (elapsed / args.steps)
        line = {
            "metric": "points/sec local-aggregation fwd+bwd (N=4096,K=32,C=64)", "value": round(value, 1),
            "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "preconditioning_steps": args.precondition,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if (args.precision == "f32" or kind != "pointwisemlp") else "bf16 contraction, f32 elsewhere",
            "data": "synthetic",
            "config": {"workload": f"ModelNet40-shape {kind} LocalAggregation fwd+bwd", "operator": kind,
                       "impl": args.impl, "clouds_per_gpu": B, "points": N, "nsample": K, "channels": C,
                       "radius": round(radius, 5), "contraction_precision": args.precision, "launch": "hip_graph" if graph is not None else "eager", "capture_stream": CAPTURE_STREAM, "optimizer": "torch.optim.SGD" if args.optimizer == "torch" else "closerlook3d_amd.optim.FlatSGD",
                       "parallelism": f"dp{world} (clouds sharded, RCCL grad all-reduce)",
                       "world_size": dist.get_world_size() if world > 1 else 1,
                       "backend": dist.get_backend() if world > 1 else None,
                       "device": f"cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}",
                       "ranks": census, "distinct_devices": len({r["device"] for r in census})},
        }
        if one_dev and world > 1:  # every rank shared device 0 over gloo: the N > 1 code path, not a scaling number
            line["config"]["one_device_standin"] = True
        if bb is not None:
            line["backbone_step"] = bb
        if not args.no_kernel_roofline:
            # top level: the TIMED STEP's dominant kernel (longest C-ABI entry point of the step table): algorithmic
            # HBM bytes per launch / median launch duration, HIP events on the launch stream; `traffic` = its PMC HBM
            # bytes from the committed passes.  `boundary` = the reference-visible ball_query + group path that
            # north_star's >= 50 % target is defined on (labelled; not what the timed step runs).
            roof = {"bound": "hbm", "peak": HBM_PEAK / 1e9, "unit": "GB/s"}
            if not args.no_step_table:
                st = step_table(compute, B, N, N, K, C, reps=20, kind=kind)
                st["graph_step_us"] = round(ms * 1e3, 1)
                # the roofline the contract asks for prices a kernel against HBM (or the matrix cores): taken for the
                # longest entry that moves data -- a VALU-bound entry (the ball query: 19 MB in ~65 us) can be the
                # longest of the table by a few microseconds and has no meaningful HBM fraction; it is named beside it
                movers = [r for r in st["kernels"] if not r["bound"].startswith("valu") and r["algorithmic_bytes"] > 0]
                d = movers[0] if movers else st["kernels"][0]
                longest = st["kernels"][0]
                if longest is not d:
                    roof["longest_entry"] = {"entry": longest["entry"], "us": longest["us"], "bound": longest["bound"],
                                             "hbm_frac": longest["hbm_frac"]}
                roof.update({"kernel": (d["kernels"] or [d["entry"]])[0], "entry": d["entry"], "us": d["us"],
                             "us_min": d["us_min"], "us_max": d["us_max"], "algorithmic_bytes": d["algorithmic_bytes"],
                             "achieved": round(d["algorithmic_bytes"] / d["us"] / 1e3, 1), "frac": d["hbm_frac"],
                             "traffic": d.get("hbm_bytes_pmc"), "l2_frac": d.get("l2_frac"),
                             "what": f"longest data-moving kernel of the timed step (fused path); bound: {d['bound']}; "
                                     "priced against HBM as the contract asks"})
            with torch.no_grad():
                per_kernel, boundary = kernel_rooflines(xyz, mask, feats.detach(), radius, K, bursts=args.bursts)
            for k, v in per_kernel.items():
                if "+" not in k:
                    v["traffic"] = pmc_traffic(k)
            roof["boundary"] = {"path": "_ext materialising (MaskedQueryAndGroup + backward as the reference's Python "
                                        "calls it; measured after the timed region -- the timed step runs the fused "
                                        "path, see 'step')",
                                "per_kernel": per_kernel, "ball_query_group": boundary}
            # the roofline fraction the headline value itself corresponds to
            roof["achieved_step"] = {"GBps": round(value / world * 17696.0 / 1e9, 1),
                                     "frac": round(value / world * 17696.0 / HBM_PEAK, 4),
                                     "definition": "points_per_s_per_gpu x 17,696 B / 8.0e12 B/s"}
            if not args.no_step_table:
                roof["step"] = st
            else:
                d = max((k for k in per_kernel if "+" not in k), key=lambda k: per_kernel[k]["ms"])
                roof.update({"kernel": d, "achieved": per_kernel[d]["achieved_GBps"], "frac": per_kernel[d]["frac"],
                             "traffic": per_kernel[d]["traffic"], "what": "boundary kernel (no step table requested)"})
            line["roofline"] = roof
            if kind == "pointwisemlp":
                line["roofline"]["contraction"] = contraction_block(B, C, N, C, args.precision)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(kind, N, K, C, radius, clouds=16, iters=2)
            line["speedup_vs_cpu_baseline"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        if bb is not None and bb.get("fatal"):  # a rank lost the others mid-step: the line is out, end the job
            sys.stdout.flush()
            os._exit(1 if rank else 0)
        dist.barrier()  # rank 0 measured the per-kernel rooflines after the timed region: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""A/B: the engine's MFMA contraction (csrc/mfma_gemm.hip, f32 and bf16) against the vendor library (torch.bmm ->
rocBLAS / hipBLASLt) on the three per-point GEMMs of a PointWiseMLP step, and MFMA utilisation against the chip peaks
(157.3 TFLOP/s f32-input MFMA, 2.5 PFLOP/s bf16 dense; MI355X_MICROARCH.md).  HIP events on the launch stream, one
launch per timing, median of `reps`.  Prints one JSON object; bench.py's `roofline.contraction` block uses
`measure()` for the shape it runs.

    python scripts/bench_point_gemm.py [--B 16 --C 64 --N 4096 --Co 64]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from closerlook3d_amd import _lib  # noqa: E402

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}
HBM_GBPS = 8000.0


def _time(fn, reps):
    for _ in range(5):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3  # seconds


def measure(B=16, C=64, N=4096, Co=64, reps=50, library=True):
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    p = lambda t: t.data_ptr()  # noqa: E731
    g = torch.Generator(device="cpu").manual_seed(0)
    f = torch.randn(B, C, N, generator=g).to(dev)
    W = (torch.randn(Co, 3 + 2 * C, generator=g) / C ** 0.5).to(dev)
    dght = torch.randn(B, N, 2 * Co, generator=g).to(dev)
    dwr = torch.randn(Co, 3, generator=g).to(dev)
    wr = torch.empty(Co, 3, device=dev)
    wcat = torch.empty(2 * Co, C, device=dev)
    ght = torch.empty(B, N, 2 * Co, device=dev)
    dfeat = torch.empty(B, C, N, device=dev)
    dW = torch.empty(Co, 3 + 2 * C, device=dev)
    ws_bytes = lib.cl3d_workspace_bytes(14, B, N, Co, 0, C)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    st = _lib.stream_ptr(dev)
    flops = 2.0 * B * N * C * 2 * Co
    io_bytes = 4.0 * (B * C * N + B * N * 2 * Co)  # one operand in, the other out (weights are negligible)
    out = {"shape": {"B": B, "C": C, "N": N, "Co": Co}, "flops_per_gemm": flops, "hbm_bytes_per_gemm": io_bytes}
    for name, prec in (("f32", 0), ("bf16", 1)):
        t_f = _time(lambda: lib.cl3d_pwmlp_point_gemm_fwd(p(f), p(W), B, C, N, Co, prec, p(ght), p(wr), p(wcat), p(ws), ws_bytes, st), reps)
        t_d = _time(lambda: lib.cl3d_pwmlp_point_gemm_bwd_data(p(dght), p(wcat), B, C, N, Co, prec, p(dfeat), p(ws), ws_bytes, st), reps)
        t_w = _time(lambda: lib.cl3d_pwmlp_point_gemm_bwd_weight(p(f), p(dght), p(dwr), B, C, N, Co, prec, p(dW), p(ws),
                                                                  ws_bytes, st), reps)
        # both gradients in one call: one kernel over d ght where it covers the shape (round 5), else the two in a row
        t_b = _time(lambda: lib.cl3d_pwmlp_point_gemm_bwd(p(f), None, None, p(dght), p(wcat), p(dwr), B, C, N, Co, prec,
                                                          p(dfeat), p(dW), p(ws), ws_bytes, st), reps)
        tot = t_f + t_d + t_w
        out["mfma_" + name] = {
            "bwd_both_us": t_b * 1e6, "bwd_both_one_kernel": bool(lib.cl3d_pwmlp_point_gemm_bwd_fused(B, C, N, Co, prec)),
            "fwd_us": t_f * 1e6, "bwd_data_us": t_d * 1e6, "bwd_weight_us": t_w * 1e6,  # each incl. its small side launch
            "tflops": 3 * flops / tot / 1e12, "frac_of_mfma_peak": 3 * flops / tot / 1e12 / PEAK_TFLOPS[name],
            "hbm_GBps": 3 * io_bytes / tot / 1e9, "frac_of_hbm_peak": 3 * io_bytes / tot / 1e9 / HBM_GBPS,
            "peak_tflops": PEAK_TFLOPS[name]}
    if library:
        wcat_t = wcat.t().unsqueeze(0).expand(B, -1, -1)
        dwb = torch.empty(B, C, 2 * Co, device=dev)

        def lib_fwd():
            lib.cl3d_pwmlp_split_weight(p(W), Co, C, p(wr), p(wcat), st)
            torch.bmm(f.transpose(1, 2), wcat_t, out=ght)

        def lib_w():
            torch.bmm(f, dght, out=dwb)
            lib.cl3d_pwmlp_merge_weight_grad(p(dwr), p(dwb), B, Co, C, p(dW), st)

        t_f = _time(lib_fwd, reps)
        t_d = _time(lambda: torch.bmm(wcat_t, dght.transpose(1, 2), out=dfeat), reps)
        t_w = _time(lib_w, reps)
        tot = t_f + t_d + t_w
        out["library_f32"] = {"fwd_us": t_f * 1e6, "bwd_data_us": t_d * 1e6, "bwd_weight_us": t_w * 1e6,
                              "tflops": 3 * flops / tot / 1e12, "frac_of_mfma_peak": 3 * flops / tot / 1e12 / PEAK_TFLOPS["f32"]}
    return out


def measure_conv(B, C, N, Co, reps=30):
    """One 1x1 convolution C -> Co over B clouds of N points: the engine's three products against torch's Conv1d
    (MIOpen / rocBLAS) forward and backward, microseconds."""
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    p = lambda t: t.data_ptr()  # noqa: E731
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(B, C, N, generator=g).to(dev)
    W = (torch.randn(Co, C, generator=g) / C ** 0.5).to(dev)
    dy = torch.randn(B, Co, N, generator=g).to(dev)
    y, dx, dW = torch.empty(B, Co, N, device=dev), torch.empty(B, C, N, device=dev), torch.empty(Co, C, device=dev)
    ws_bytes = lib.cl3d_workspace_bytes(15, B, N, Co, 0, C)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    st = _lib.stream_ptr(dev)
    out = {"conv": {"B": B, "C": C, "N": N, "Co": Co}, "gflop": 2.0 * B * N * C * Co / 1e9}
    for name, prec in (("f32", 0), ("bf16", 1)):
        out[name] = [round(_time(fn, reps) * 1e6, 1) for fn in (
            lambda: lib.cl3d_conv1x1_fwd(p(x), p(W), B, C, N, Co, prec, p(y), p(ws), ws_bytes, st),
            lambda: lib.cl3d_conv1x1_bwd_data(p(dy), p(W), B, C, N, Co, prec, p(dx), p(ws), ws_bytes, st),
            lambda: lib.cl3d_conv1x1_bwd_weight(p(x), p(dy), B, C, N, Co, prec, p(dW), p(ws), ws_bytes, st))]
    conv = torch.nn.Conv1d(C, Co, 1, bias=False).to(dev)
    xr = x.clone().requires_grad_(True)
    t_f = _time(lambda: conv(xr), reps)

    def fb():
        conv.weight.grad = None
        xr.grad = None
        conv(xr).backward(dy)

    t_fb = _time(fb, reps)
    out["library_f32"] = [round(t_f * 1e6, 1), round((t_fb - t_f) * 1e6, 1)]  # forward, backward (both gradients)
    return out


CONFIG2_CONVS = [(3, 72, 4096), (72, 72, 4096), (72, 144, 4096), (144, 144, 4096), (144, 288, 1024), (288, 144, 1024),
                 (288, 288, 1024), (288, 576, 256), (576, 288, 256), (576, 576, 256), (576, 1152, 64), (1152, 576, 64),
                 (1152, 1152, 64), (1152, 2304, 16), (2304, 1152, 16)]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--C", type=int, default=64)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--Co", type=int, default=64)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--sweep", action="store_true", help="also the ModelNet backbone's operator shapes (width 144)")
    ap.add_argument("--convs", action="store_true", help="the 1x1 convolutions of the ModelNet backbone (config 2) vs torch Conv1d")
    a = ap.parse_args()
    if a.convs:
        for C, Co, N in CONFIG2_CONVS:
            print(json.dumps(measure_conv(a.B, C, N, Co, a.reps)))
        sys.exit(0)
    res = [measure(a.B, a.C, a.N, a.Co, a.reps)]
    if a.sweep:
        for C, N in ((72, 4096), (144, 1024), (288, 256), (576, 64)):
            res.append(measure(16, C, N, C, a.reps))
    for r in res:
        print(json.dumps(r))

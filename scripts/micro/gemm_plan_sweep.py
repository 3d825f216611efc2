"""Tile / K-split sweep of the 1x1-convolution products of the config-2 backbone (csrc/mfma_gemm.hip), the data the
planner's cost model (plan_gemm) is fitted to.  Needs a variant library built with -DCL3D_GEMM_PLAN_ENV (the shipped
library has no override):

  python scripts/micro/gemm_plan_sweep.py --build     (build container)
  python scripts/micro/gemm_plan_sweep.py --run       (GPU box; one JSON line per (layer, product, tile, split))

Timing: `reps` launches captured into one HIP graph, replayed between two HIP events; per launch = total / reps.
"""
import argparse
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "closerlook3d_amd", "csrc")
VAR = os.path.join(ROOT, "scripts", "micro", "var")
LIB = os.path.join(VAR, "libcl3d_gemm_plan_env.so")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-fno-slp-vectorize",
         "-DCL3D_D2_FORM=0", "-DCL3D_GEMM_PLAN_ENV"]

LAYERS = [  # C, points per cloud, Co  (B = 16 clouds)
    (72, 4096, 144), (144, 4096, 144), (144, 1024, 288), (288, 1024, 144), (288, 1024, 288),
    (288, 256, 576), (576, 256, 288), (576, 256, 576), (576, 64, 1152), (1152, 64, 576), (1152, 64, 1152),
    (1152, 16, 2304), (2304, 16, 1152), (1152, 16, 1152),
]


def build():
    os.makedirs(VAR, exist_ok=True)
    objs = [o for o in sorted(glob.glob(os.path.join(CSRC, "*.o"))) if ".d2form" not in o and not o.endswith("mfma_gemm.o")]
    obj = os.path.join(VAR, "var_gemm_plan_env.o")
    subprocess.check_call([HIPCC] + FLAGS + ["-c", os.path.join(CSRC, "mfma_gemm.hip"), "-o", obj])
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [obj])
    os.remove(obj)
    print("built", LIB)


def run(reps, precisions, point_products=False, fused_sum="1"):
    os.environ["CL3D_LIB"] = LIB
    os.environ["CL3D_GEMM_FUSED_SUM"] = fused_sum  # 1: K slices summed inside the launch (round 6), 0: by a launch of their own
    import torch
    sys.path.insert(0, ROOT)
    from closerlook3d_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    p = lambda t: t.data_ptr()  # noqa: E731
    B = 16
    ws = torch.empty(512 << 20, dtype=torch.uint8, device=dev)

    def timed(fn):
        # `reps` launches captured into one HIP graph and replayed: kernel time + the ~1 us between graph nodes, without
        # the host's enqueue rate (a ctypes call + hipLaunchKernel is ~15 us: launched eagerly, every kernel shorter than
        # that reads 15-20 us)
        fn(_lib.stream_ptr(dev))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            st = _lib.stream_ptr(dev)
            for _ in range(reps):
                fn(st)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        g.replay()
        for _ in range(3):
            a.record()
            g.replay()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / reps)
        return best

    if point_products:
        layers = [(72, 4096, 72), (36, 4096, 36), (72, 1024, 72), (144, 1024, 144), (144, 256, 144), (288, 256, 288),
                  (288, 64, 288), (576, 64, 576), (576, 16, 576), (1152, 16, 1152)]
    else:
        layers = LAYERS
    for C, n, Co in layers:
        x = torch.randn(B, C, n, device=dev)
        dy = torch.randn(B, Co, n, device=dev)
        W = torch.randn(Co, C, device=dev) / C ** 0.5
        y = torch.empty(B, Co, n, device=dev)
        dx = torch.empty(B, C, n, device=dev)
        dW = torch.empty(Co, C, device=dev)
        for prec_name in precisions:
            prec = 1 if prec_name == "bf16" else 0
            products = {
                "fwd": lambda st: lib.cl3d_conv1x1_fwd(p(x), p(W), B, C, n, Co, prec, p(y), p(ws), ws.numel(), st),
                "bwd_data": lambda st: lib.cl3d_conv1x1_bwd_data(p(dy), p(W), B, C, n, Co, prec, p(dx), p(ws), ws.numel(), st),
                "bwd_weight": lambda st: lib.cl3d_conv1x1_bwd_weight(p(x), p(dy), B, C, n, Co, prec, p(dW), p(ws), ws.numel(), st),
            }
            if point_products:  # the PointWiseMLP's per-point contraction [B n, C] x [C, 2 Co] and its two gradients
                Wp = torch.randn(Co, 3 + 2 * C, device=dev) / C ** 0.5
                ght = torch.empty(B, n, 2 * Co, device=dev)
                dght = torch.randn(B, n, 2 * Co, device=dev)
                wr, wcat = torch.empty(Co, 3, device=dev), torch.empty(2 * Co, C, device=dev)
                dwr, dWp = torch.randn(Co, 3, device=dev), torch.empty(Co, 3 + 2 * C, device=dev)
                lib.cl3d_pwmlp_point_gemm_fwd(p(x), p(Wp), B, C, n, Co, prec, p(ght), p(wr), p(wcat), p(ws), ws.numel(), _lib.stream_ptr(dev))
                products = {
                    "pt_fwd": lambda st: lib.cl3d_pwmlp_point_gemm_fwd(p(x), p(Wp), B, C, n, Co, prec, p(ght), p(wr), p(wcat), p(ws), ws.numel(), st),
                    "pt_bwd_data": lambda st: lib.cl3d_pwmlp_point_gemm_bwd_data(p(dght), p(wcat), B, C, n, Co, prec, p(dx), p(ws), ws.numel(), st),
                    "pt_bwd_weight": lambda st: lib.cl3d_pwmlp_point_gemm_bwd_weight(p(x), p(dght), p(dwr), B, C, n, Co, prec, p(dWp), p(ws), ws.numel(), st),
                }
            for name, fn in products.items():
                os.environ.pop("CL3D_GEMM_FORCE", None)
                auto = timed(fn)
                print(json.dumps({"layer": [C, n, Co], "prec": prec_name, "product": name, "plan": "auto", "fused_sum": fused_sum, "us": round(auto, 2)}), flush=True)
                splits = [1, 2, 3, 4, 6, 8, 12, 16] if not name.endswith("bwd_weight") else [1, 2, 4, 8, 16, 32, 64, 128, 256]
                for wi, wj in ((1, 1), (2, 1), (1, 2), (2, 2)):
                    row = {}
                    for sp in splits:
                        os.environ["CL3D_GEMM_FORCE"] = f"{wi},{wj},{sp}"
                        row[sp] = round(timed(fn), 2)
                    print(json.dumps({"layer": [C, n, Co], "prec": prec_name, "product": name, "tile": [wi, wj], "fused_sum": fused_sum, "us_by_split": row}), flush=True)
    os.environ.pop("CL3D_GEMM_FORCE", None)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--precisions", default="bf16")
    ap.add_argument("--point", action="store_true", help="the PointWiseMLP per-point products instead of the convolutions")
    ap.add_argument("--fused-sum", default="1", choices=["0", "1"], help="how K slices are summed (1 = inside the launch)")
    a = ap.parse_args()
    if a.build:
        build()
    if a.run:
        run(a.reps, a.precisions.split(","), a.point, a.fused_sum)

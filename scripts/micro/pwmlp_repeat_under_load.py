"""The PointWiseMLP's forward pieces, launched REPS times on the same operands beside a second process doing the same: which
intermediate varies?  (Round 6, session 44: the first varying output of the bf16 two-rank step was _PointwiseMLP's, its inputs did not.)"""
import os, sys, subprocess, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import closerlook3d_amd  # noqa: E402
from closerlook3d_amd import fused, pt_utils, _lib  # noqa: E402

REPS = int(os.environ.get("REPS", "150"))
dev = torch.device("cuda:0")
_p, _stream = fused._p, fused._stream


def setup(C=72, N=4096, B=16, K=32, radius=0.14, seed=3):
    g = torch.Generator(device="cpu").manual_seed(seed)
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    mask = torch.ones(B, N, dtype=torch.int32).to(dev)
    feats = torch.randn(B, C, N, generator=g).to(dev)
    W = (torch.randn(C, 3 + 2 * C, generator=g) / C ** 0.5).to(dev)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    gamma[::5] *= -1
    beta = torch.randn(C, generator=g).to(dev)
    return xyz, mask, feats, W, gamma, beta, K, radius


def forward_pieces(xyz, mask, feats, W, gamma, beta, K, radius, prec):
    lib = _lib.lib()
    B, C, N = feats.shape
    Co = W.shape[0]
    idx, _ = pt_utils._ball_query(xyz, xyz, mask, mask, radius, K)
    ght, wr = fused._PointRows.apply(feats, W, prec)
    ght = ght.contiguous()
    M = N
    nparts = lib.cl3d_pwmlp_partials(B, M, Co)
    vec = torch.empty((4, Co), dtype=torch.float32, device=dev)
    ystar = torch.empty((B, M, Co), dtype=torch.float32, device=dev)
    sy = torch.empty((B, M, Co), dtype=torch.float32, device=dev)
    kstar = torch.empty((B, M, Co), dtype=torch.uint8, device=dev)
    partial = torch.empty((nparts, Co, 8), dtype=torch.float64, device=dev)
    sums = torch.empty((Co, 6), dtype=torch.float64, device=dev)
    rm, rv = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    out = torch.empty((B, Co, M), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        st = _stream(ght)
        _lib.check(lib.cl3d_pwmlp_stats(_p(xyz), _p(xyz), _p(idx), _p(ght), _p(wr), _p(gamma), B, N, M, K, Co, float(radius),
                                        _p(ystar), _p(kstar), _p(sy), _p(partial), nparts, st))
        _lib.check(lib.cl3d_pwmlp_finalize_stats(_p(partial), nparts, Co, float(B * M * K), 1e-5, 0.1, _p(gamma), _p(beta), _p(rm), _p(rv),
                                                 None, _p(vec[0]), _p(vec[1]), _p(vec[2]), _p(vec[3]), _p(sums), st))
        _lib.check(lib.cl3d_pwmlp_apply(_p(ystar), _p(vec[0]), _p(vec[1]), B, M, Co, _p(out), st))
    return {"idx": idx, "ght": ght, "wr": wr, "ystar": ystar, "kstar": kstar.int(), "sy": sy,
            "partial": partial.view(torch.int32), "vec": vec, "sums": sums.view(torch.int32), "out": out}


def bits(t):
    return int(t.contiguous().view(torch.int32).long().sum()) if t.dtype in (torch.float32, torch.int32) else int(t.long().sum())


def sweep(tag):
    for C, N in ((144, 4096), (72, 4096), (144, 1024))[:int(os.environ.get("VICTIMS", "3"))]:
        args = setup(C=C, N=N, radius=0.14 if N == 4096 else 0.22)
        for pname, prec in (("f32", 0), ("bf16", 1))[:int(os.environ.get("VICTIM_PRECS", "2"))]:
            seen = {}
            for _ in range(REPS):
                o = forward_pieces(*args, prec)
                torch.cuda.synchronize()
                for k, v in o.items():
                    seen.setdefault(k, set()).add(bits(v))
            print("%-26s C=%d N=%d %-5s distinct bit patterns over %d launches: %s" % (tag, C, N, pname, REPS, {k: len(v) for k, v in seen.items()}), flush=True)


def load_loop(seconds):
    t0 = time.time()
    args = setup()
    print("load: running", flush=True)
    while time.time() - t0 < seconds:
        for _ in range(20):
            forward_pieces(*args, 1)
        torch.cuda.synchronize()


def gemm_load(spec, seconds=120.0):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bf16_repeat_under_load as g
    _, prec, kind, first, last = spec.split("_")
    cs = [c for c in g.cases() if kind in ("all", c[1])][int(first):int(last)]
    print("load:", [c[0] for c in cs], flush=True)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for name, k, x, W, dy in cs:
            g.one(k, x, W, dy, 1 if prec == "bf16" else 0)
        torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--gemm-load":
        gemm_load(sys.argv[2])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--load":
        load_loop(float(sys.argv[2]))
        sys.exit(0)
    if os.environ.get("SKIP_ALONE") != "1":
        sweep("alone")
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
    for load in os.environ.get("LOADS", "self,bb_bf16,bb_f32").split(","):
        if load.startswith("gemm"):  # gemm_<prec>_<kind>_<first>_<last>: the contractions of bf16_repeat_under_load.py in a loop
            child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--gemm-load", load], stdout=subprocess.PIPE, text=True)
            child.stdout.readline()
        elif load == "self":
            child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--load", "120"], stdout=subprocess.PIPE, text=True)
            child.stdout.readline()  # "load: running"
        else:  # a whole config-2 backbone step in a loop (what the second rank of the one-device stand-in is)
            child = subprocess.Popen([sys.executable, os.path.join(root, "scripts", "bench_backbone.py"), "--config", "modelnet_pointwisemlp",
                                      "--precision", load[3:], "--steps", "4000", "--gemm-plans", "model"] + (["--no-graph"] if os.environ.get("LOAD_EAGER") == "1" else []),
                                     stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            time.sleep(14.0)  # (import, warm-up, capture)
        time.sleep(1.0)
        try:
            sweep("beside %s" % load)
        finally:
            child.terminate()
            child.wait()

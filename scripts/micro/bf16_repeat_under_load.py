"""Is a contraction's output bit-repeatable while ANOTHER process keeps the GPU busy?  (Round 6, sessions 37-42: the bf16 step of two
ranks on one device varied from replay to replay, f32 did not.)  Each product is launched REPS times on the same operands and the
distinct output bit patterns are counted, first alone, then beside a child process that loops over the same products.
    python scripts/micro/bf16_repeat_under_load.py            # both precisions, alone and under load
"""
import os, sys, subprocess, time, json
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import closerlook3d_amd  # noqa: E402
from closerlook3d_amd import fused  # noqa: E402

REPS = int(os.environ.get("REPS", "40"))
dev = torch.device("cuda:0")


def cases():
    g = torch.Generator(device="cpu").manual_seed(1)
    B = 16
    for C, Co, N in [(3, 72, 4096), (72, 72, 4096), (72, 36, 4096), (36, 144, 4096), (144, 72, 1024), (72, 288, 1024), (288, 144, 256),
                     (576, 288, 64), (1152, 576, 16), (576, 2304, 16)]:
        x = torch.randn(B, C, N, generator=g).to(dev)
        W = (torch.randn(Co, C, generator=g) / C ** 0.5).to(dev)
        dy = torch.randn(B, Co, N, generator=g).to(dev)
        yield ("conv1x1 %d->%d N=%d" % (C, Co, N), "conv", x, W, dy)
    for C, N in [(36, 4096), (72, 4096), (64, 4096), (144, 1024), (288, 256), (576, 64), (1152, 16)]:
        x = torch.randn(B, C, N, generator=g).to(dev)
        W = (torch.randn(C, 3 + 2 * C, generator=g) / C ** 0.5).to(dev)
        dy = torch.randn(B, N, 2 * C, generator=g).to(dev)
        yield ("point rows C=%d N=%d" % (C, N), "rows", x, W, dy)


def one(kind, x, W, dy, prec):
    x = x.clone().requires_grad_(True)
    W = W.clone().requires_grad_(True)
    if kind == "conv":
        y = fused._Conv1x1.apply(x, W, prec)
        y.backward(dy)
        return [y.detach(), x.grad, W.grad]
    ght, wr = fused._PointRows.apply(x, W, prec)
    (ght * dy).sum().backward()
    return [ght.detach(), x.grad, W.grad]


def bits(t):
    return int(t.contiguous().view(torch.int32).long().sum())


def sweep(tag):
    rows = []
    for name, kind, x, W, dy in cases():
        for pname, prec in (("f32", 0), ("bf16", 1)):
            seen = [set(), set(), set()]
            for _ in range(REPS):
                outs = one(kind, x, W, dy, prec)
                torch.cuda.synchronize()
                for s, o in zip(seen, outs):
                    s.add(bits(o))
            rows.append((name, pname, [len(s) for s in seen]))
    bad = [r for r in rows if max(r[2]) > 1]
    print("== %s: %d products x %d launches, %d with more than one bit pattern [fwd, d x, d W]" % (tag, len(rows), REPS, len(bad)), flush=True)
    for r in bad:
        print("   %-28s %-5s %s" % r, flush=True)
    return bad


def load_loop(seconds):
    t0 = time.time()
    cs = list(cases())
    while time.time() - t0 < seconds:
        for name, kind, x, W, dy in cs:
            one(kind, x, W, dy, 1)
        torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--load":
        load_loop(float(sys.argv[2]))
        sys.exit(0)
    sweep("alone")
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--load", os.environ.get("LOAD_S", "60")])
    time.sleep(8)  # (the child's import and first launches)
    try:
        sweep("beside a second process")
    finally:
        child.terminate()
        child.wait()

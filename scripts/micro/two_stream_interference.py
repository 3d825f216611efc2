"""ONE process, two streams: bf16 (or f32) contractions in a loop on stream A, the PointWiseMLP gather pass on stream B -- do the gather
pass's outputs vary from launch to launch?  (Round 6, sessions 50-51: they do beside a SECOND PROCESS running bf16 contractions.)"""
import os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pwmlp_repeat_under_load as v  # noqa: E402
import bf16_repeat_under_load as g  # noqa: E402

dev = torch.device("cuda:0")
REPS = int(os.environ.get("REPS", "120"))


def run(prec_name):
    prec = 1 if prec_name.endswith("bf16") else 0
    cs = list(g.cases())[5:10]
    stop = [False]
    sa = torch.cuda.Stream()

    ta = torch.randn(4096, 2304, device=dev, dtype=torch.bfloat16 if "bf16" in prec_name else torch.float32)
    tb = torch.randn(2304, 1152, device=dev, dtype=ta.dtype)

    def loader():
        torch.cuda.set_device(0)
        with torch.cuda.stream(sa):
            while not stop[0]:
                if prec_name.startswith("torch"):  # the vendor library's contraction instead of the engine's
                    for _ in range(20):
                        torch.matmul(ta, tb)
                else:
                    for name, k, x, W, dy in cs:
                        g.one(k, x, W, dy, prec)
                sa.synchronize()
    th = threading.Thread(target=loader)
    th.start()
    time.sleep(0.5)
    args = v.setup(C=144, N=4096, radius=0.14)
    seen = {}
    sb = torch.cuda.Stream()
    with torch.cuda.stream(sb):
        for _ in range(REPS):
            o = v.forward_pieces(*args, 0)
            sb.synchronize()
            for k_, t in o.items():
                seen.setdefault(k_, set()).add(v.bits(t))
    stop[0] = True
    th.join()
    print("one process, %s contractions on a second stream (torch_* = torch.matmul): distinct bit patterns over %d launches: %s" % (prec_name, REPS, {k_: len(s) for k_, s in seen.items()}), flush=True)


if __name__ == "__main__":
    for name in os.environ.get("RUNS", "f32,bf16,torch_f32,torch_bf16,bf16,torch_bf16").split(","):
        run(name)

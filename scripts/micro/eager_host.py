"""Where the eager (no HIP graph) bench step spends its HOST time: enqueue time of the step against its synchronized
time, with pieces removed one at a time, and a cProfile of the full step.  python scripts/micro/eager_host.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import make_config, synth_batch  # noqa: E402
import numpy as np  # noqa: E402


def main():
    from closerlook3d_amd.local_aggregation_operators import LocalAggregation
    dev = torch.device("cuda", 0)
    B, N, K, C = 16, 4096, 32, 64
    radius = float((1.5 * K * 3 / (4 * np.pi * N)) ** (1 / 3))
    torch.manual_seed(0)
    module = LocalAggregation(C, C, radius, K, make_config("pointwisemlp", "auto")).to(dev).train(True)
    params = [p for p in module.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3)
    xyz, mask, feats = (torch.from_numpy(a).to(dev) for a in synth_batch(B, N, C, 1000))
    feats.requires_grad_(True)
    probe = torch.randn(B, C, N, device=dev)

    def full():
        feats.grad = None
        opt.zero_grad(set_to_none=True)
        out = module(xyz, xyz, mask, mask, feats)
        out.backward(probe)
        opt.step()

    def no_opt():
        feats.grad = None
        for p in params:
            p.grad = None
        out = module(xyz, xyz, mask, mask, feats)
        out.backward(probe)

    def fwd_only():
        module(xyz, xyz, mask, mask, feats)

    for name, fn in (("full step", full), ("without the optimizer", no_opt), ("forward only", fwd_only)):
        for _ in range(300):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: host enqueue {1e3 * (t1 - t0) / 500:.4f} ms/step, with the device drained {1e3 * (t2 - t0) / 500:.4f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        full()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()

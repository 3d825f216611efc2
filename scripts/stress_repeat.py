"""Run-to-run repeatability of an operator's forward/backward at the benchmark shape (debug aid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_config, synth_batch
from closerlook3d_amd.local_aggregation_operators import LocalAggregation

kind, impl, iters = sys.argv[1], sys.argv[2], int(sys.argv[3])
B, N, K, C = 16, 4096, 32, 64
radius = float((1.5 * K * 3 / (4 * np.pi * N)) ** (1 / 3))
xyz, mask, feats = (torch.from_numpy(a).cuda() for a in synth_batch(B, N, C, 5))
probe = torch.randn(B, C, N, device="cuda")
torch.manual_seed(3)
mod = LocalAggregation(C, C, radius, K, make_config(kind, impl)).cuda().train(True)
ref = None
worst = [0.0, 0.0]
for it in range(iters):
    f = feats.clone().requires_grad_(True)
    mod.zero_grad()
    out = mod(xyz, xyz, mask, mask, f)
    (out * probe).sum().backward()
    cur = (out.detach().clone(), f.grad.clone())
    if ref is None:
        ref = cur
        continue
    for q in range(2):
        rel = ((cur[q] - ref[q]).double().norm() / ref[q].double().norm()).item()
        worst[q] = max(worst[q], rel)
        if rel > 1e-6:
            d = (cur[q] - ref[q]).abs()
            idx = np.unravel_index(int(d.argmax()), d.shape)
            print(f"iter {it}: {'out' if q == 0 else 'grad'} rel {rel:.2e}, max |diff| {d.max().item():.3e} at {idx}, "
                  f"#elements off by > 1e-4: {(d > 1e-4).sum().item()}")
print(kind, impl, "worst rel (out, grad):", worst)

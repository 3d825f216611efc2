"""Phase timing of grid_subsample_kernel by early exit: libraries built with -DCL3D_SUB_PHASE=n return after phase n
(1 bounding box, 6 keys made but not sorted, 2 keys + sort, 3 cell heads, 4 shuffle tables, 5 barycentres; 0 = the shipped
kernel).  Build the variants with  for n in 1..6: hipcc <library flags> -DCL3D_SUB_PHASE=n -c csrc/grid_subsample.hip,  linked with
the library's other objects into scripts/micro/var/libcl3d_sub_phase<n>.so.
    python scripts/micro/grid_subsample_phases.py       (GPU box; variants under scripts/micro/var/)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import torch, sys
sys.path.insert(0, %r)
from closerlook3d_amd import _ext
dev = torch.device('cuda:0')
out = {}
for N in (4096, 1024, 256, 64):
    B = 16
    g = torch.Generator(device='cpu').manual_seed(N)
    pts = torch.rand(B, N, 3, generator=g).to(dev); mask = torch.ones(B, N, dtype=torch.int32, device=dev)
    dl = 0.03 * (4096 / N) ** (1 / 3)
    fn = lambda: _ext.masked_grid_subsampling(pts, mask, N // 4, dl)
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr.replay(); best = 1e9
    for _ in range(3):
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) * 100)
    out[N] = round(best, 1)
import json; print(json.dumps(out))
''' % ROOT

for ph in (1, 6, 2, 3, 4, 5, 0):
    env = dict(os.environ)
    if ph:
        env["CL3D_LIB"] = os.path.join(ROOT, "scripts", "micro", "var", f"libcl3d_sub_phase{ph}.so")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print(json.dumps({"returns_after_phase": ph, "us_by_N": json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else r.stderr[-300:]}), flush=True)

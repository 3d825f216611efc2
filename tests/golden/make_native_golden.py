"""Generate tests/golden/native_*.npz from the REFERENCE's own kernels (oracle/_ref/ref_ext.so).

Runs on the MI355X box (the reference's five ops have no CPU path):
    python tests/golden/make_native_golden.py gpurun_out/native_golden
then the .npz files are copied into tests/golden/ and committed.  Each file: seeded inputs and what
the reference's compiled extension returned for all five ops on them.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402
from oracle import operators as oo  # noqa: E402

CASES = [
    # name, B, N, M, K, in-radius multiple, kind, pad, npoint, sampleDl, C
    ("small_uniform", 2, 256, 256, 16, 1.5, "uniform", 0.25, 64, 0.12, 6),
    ("dense_planes", 2, 1024, 1024, 16, 5.0, "planes", 0.1, 256, 0.08, 4),
    ("ragged", 3, 777, 130, 7, 2.0, "uniform", 0.3, 100, 0.1, 5),
    ("metric_shape", 1, 4096, 4096, 32, 1.5, "uniform", 0.0, 1024, 0.04, 3),
]


def main(out_dir):
    ref = build_ref.load()
    assert ref is not None, "oracle/_ref/ref_ext.so missing (python oracle/build_ref.py in the build container)"
    os.makedirs(out_dir, exist_ok=True)
    dev = torch.device("cuda:0")
    for ci, (name, B, N, M, K, mult, kind, pad, npoint, dl, C) in enumerate(CASES):
        rng = np.random.default_rng(7000 + ci)
        s, sm = oo.make_cloud(rng, B, N, kind=kind, pad_frac=pad)
        if M == N:
            q, qm = s.copy(), sm.copy()
        else:
            sel = rng.integers(0, N, (B, M))
            q = (np.take_along_axis(s, sel[..., None], 1) + 0.004 * rng.standard_normal((B, M, 3))).astype(np.float32)
            qm = np.ones((B, M), np.int32)
            qm[:, M - M // 5:] = 0
        radius = float((mult * K * 3 / (4 * np.pi * N)) ** (1 / 3))
        feats = rng.standard_normal((B, C, N)).astype(np.float32)
        grad_out = rng.standard_normal((B, C, M, K)).astype(np.float32)
        t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
        idx, idx_mask = ref.masked_ordered_ball_query(t(q), t(s), t(qm), t(sm), radius, K)
        nidx, nmask = ref.masked_nearest_query(t(q), t(s), t(qm), t(sm))
        sub, smask = ref.masked_grid_subsampling(t(s), t(sm), npoint, dl)
        grouped = ref.group_points(t(feats), idx)
        gp = ref.group_points_grad(t(grad_out), idx, N)
        torch.cuda.synchronize()
        # queries with no in-radius support hit `i % 0` in the reference: mark them undefined
        d2 = ((q[:, :, None, :].astype(np.float64) - s[:, None, :, :]) ** 2).sum(-1)
        nv = np.where((sm == 0).any(1), (sm == 0).argmax(1), N)
        valid = np.arange(N)[None, None, :] < nv[:, None, None]
        defined = ((d2 < (radius * 0.999) ** 2) & valid).any(-1)
        np.savez_compressed(
            os.path.join(out_dir, f"native_{name}.npz"), query_xyz=q, support_xyz=s, query_mask=qm, support_mask=sm,
            radius=np.float32(radius), nsample=np.int32(K), npoint=np.int32(npoint), sampleDl=np.float32(dl),
            features=feats, grad_out=grad_out, bq_idx=idx.cpu().numpy(), bq_idx_mask=idx_mask.cpu().numpy(),
            bq_defined=defined, nn_idx=nidx.cpu().numpy(), nn_idx_mask=nmask.cpu().numpy(),
            sub_xyz=sub.cpu().numpy(), sub_mask=smask.cpu().numpy(), grouped=grouped.cpu().numpy(),
            grad_points=gp.cpu().numpy())
        print(name, "ok", idx.shape, int(idx_mask.sum()))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "native_golden"))

"""Randomised pin of the oracle's operator restatement (oracle/operators.py) against the REFERENCE's own
`LocalAggregation` module, beyond the committed fixtures: random operator variants, shapes, neighbour counts,
radii, padding, cloud geometry, train / eval.  Runs only in the build container (imports /root/reference; the
native ops under the reference's Python are the oracle's C restatement, as in make_operator_golden.py).

    python tests/golden/fuzz_oracle_vs_reference.py [trials] [seed]   ->  tests/golden/fuzz_oracle_vs_reference.log
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_operator_golden as mog  # noqa: E402
from tests.helpers import assert_close, oracle_operator  # noqa: E402

VARIANTS = [
    ("pospool", lambda r: dict(pospool__position_embedding=r.choice(["xyz", "sin_cos"]),
                               pospool__reduction=r.choice(["avg", "sum", "max"]),
                               pospool__output_conv=bool(r.integers(0, 2)))),
    ("adaptive_weight", lambda r: dict(adaptive_weight__num_mlps=int(r.integers(1, 4)),
                                       adaptive_weight__shared_channels=int(r.choice([1, 2, 3, 6])),
                                       adaptive_weight__reduction=r.choice(["avg", "sum", "max"]),
                                       adaptive_weight__output_conv=bool(r.integers(0, 2)))),
    ("pointwisemlp", lambda r: dict(pointwisemlp__feature_type="dp_fi_df", pointwisemlp__num_mlps=int(r.integers(1, 4)),
                                    pointwisemlp__reduction=r.choice(["max", "avg", "sum"]))),
    ("pseudo_grid", lambda r: dict(pseudo_grid__KP_influence=r.choice(["linear", "constant"]),
                                   pseudo_grid__output_conv=bool(r.integers(0, 2)))),
]


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
    os.environ["JOB_LOG_DIR"] = tempfile.mkdtemp(prefix="cl3d_fuzz_")
    mog._install_stubs()
    sys.path.insert(0, mog.REF)
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("gloo", rank=0, world_size=1)
    from models.local_aggregation_operators import LocalAggregation
    rng = np.random.default_rng(seed0)
    lines, worst, t0 = [], {}, time.time()
    skipped = 0
    for t in range(trials):
        kind, make = VARIANTS[t % len(VARIANTS)]
        over = {k: (v.item() if hasattr(v, "item") else v) for k, v in make(rng).items()}
        over = {k: (str(v) if isinstance(v, np.str_) else v) for k, v in over.items()}
        B, N, K = int(rng.integers(1, 4)), int(rng.integers(48, 320)), int(rng.integers(3, 25))
        C = 6 * int(rng.integers(1, 5))
        mult = float(rng.choice([0.6, 1.5, 4.0]))
        pad = float(rng.choice([0.0, 0.2, 0.5]))
        geom = str(rng.choice(["uniform", "planes"]))
        train = bool(rng.integers(0, 4) > 0)
        cfg = mog._config(kind, **over)
        radius = mog._radius(N, K, mult)
        torch.manual_seed(seed0 + t)
        np.random.seed(seed0 + t)
        try:
            mod = LocalAggregation(C, C, radius, K, cfg)
        except Exception as e:  # a combination the reference itself rejects (e.g. shared_channels not dividing C)
            skipped += 1
            lines.append(f"{t:4d} {kind:16s} skipped: reference constructor raised {type(e).__name__}")
            continue
        mog._randomize_bn(mod, seed0 + 1000 + t)
        mod.train(train)
        xyz, mask, feats = mog._inputs(seed0 + 2000 + t, B, N, C, pad, kind=geom)
        state = mog._state(mod)
        try:
            rec = mog._run_module(mod, [torch.from_numpy(a) for a in (xyz, xyz, mask, mask, feats)], 4, seed0 + 3000 + t)
        except RuntimeError as e:  # the reference's own forward/backward fails for this variant
            skipped += 1
            lines.append(f"{t:4d} {kind:16s} skipped: the reference raised RuntimeError for {over}: {str(e)[:90]}")
            continue
        fx = dict(rec)
        fx.update(state)
        fx.update(xyz=xyz, mask=mask, features=feats, radius=np.float32(radius), nsample=np.int32(K),
                  training=np.int32(train), kind=kind, over=over)
        y, gf, grads = oracle_operator(fx)
        tag = f"{t:4d} {kind:16s} B={B} N={N:3d} K={K:2d} C={C:2d} r*={mult} pad={pad} {geom:7s} {'train' if train else 'eval '} {over}"
        assert_close(y.numpy(), fx["out"], 1e-5, tag + ": out")
        assert_close(gf.numpy(), fx["grad_features"], 2e-5, tag + ": grad_features")
        err = float(np.abs(y.numpy() - fx["out"]).max())
        for k, g in grads.items():
            if "grad__" + k in fx:
                assert_close(g.numpy(), fx["grad__" + k], 5e-5, tag + f": grad {k}")
        worst[kind] = max(worst.get(kind, 0.0), err)
        lines.append(tag + f"  max|out err| {err:.2e}")
    summary = (f"{trials - skipped} random cases agree with the reference's LocalAggregation (out 1e-5, d features 2e-5, "
               f"d parameters 5e-5; {skipped} combinations on which the reference's own constructor or backward raises), seed {seed0}, "
               f"{time.time() - t0:.0f} s; worst |out error| per operator: "
               + ", ".join(f"{k} {v:.1e}" for k, v in sorted(worst.items())))
    print(summary)
    with open(os.path.join(ROOT, "tests", "golden", "fuzz_oracle_vs_reference.log"), "w") as fh:
        fh.write(summary + "\n" + "\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

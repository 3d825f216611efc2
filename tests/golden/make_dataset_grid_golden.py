"""tests/golden/dataset_grid_*.npz: inputs and the REFERENCE's outputs of the dataset-side grid subsampling
(grid_subsampling.cpp compiled as is, oracle/build_ref.py:build_grid -> oracle/_ref/libgrid_dataset_ref.so).
Build container only.    python tests/golden/make_dataset_grid_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402


def run_ref(lib, p, f, l, dl):
    n = len(p)
    fdim = 0 if f is None else f.shape[1]
    ldim = 0 if l is None else l.shape[1]
    sp = np.empty((n, 3), np.float32)
    sf = np.empty((n, max(fdim, 1)), np.float32)
    sl = np.empty((n, max(ldim, 1)), np.int32)
    m = lib.cl3d_ref_dataset_grid_subsampling(p.ctypes.data, None if f is None else f.ctypes.data,
                                              None if l is None else l.ctypes.data, n, fdim, ldim, dl,
                                              sp.ctypes.data, sf.ctypes.data, sl.ctypes.data)
    return sp[:m].copy(), sf[:m, :fdim].copy(), sl[:m, :ldim].copy()


def main():
    build_ref.build_grid()
    lib = build_ref.load_grid()
    rng = np.random.default_rng(2024)
    cases = {
        "room": (4000, 0.25, 4, 1, lambda n: rng.random((n, 3), dtype=np.float32) * np.float32([6, 4, 3])),
        "negative": (3000, 0.1, 3, 2, lambda n: (rng.random((n, 3), dtype=np.float32) - 0.7).astype(np.float32) * 2),
        "points_only": (2500, 0.05, 0, 0, lambda n: rng.standard_normal((n, 3)).astype(np.float32) * 0.3),
        "labels_only": (2000, 0.3, 0, 1, lambda n: rng.random((n, 3), dtype=np.float32) * 2),
    }
    for name, (n, dl, fdim, ldim, gen) in cases.items():
        p = np.ascontiguousarray(gen(n), np.float32)
        f = rng.random((n, fdim), dtype=np.float32) if fdim else None
        l = rng.integers(0, 13, (n, ldim)).astype(np.int32) if ldim else None
        sp, sf, sl = run_ref(lib, p, f, l, dl)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"dataset_grid_{name}.npz"), points=p,
                            features=f if f is not None else np.zeros((0, 0), np.float32),
                            labels=l if l is not None else np.zeros((0, 0), np.int32), sampleDl=np.float32(dl),
                            ref_points=sp, ref_features=sf, ref_labels=sl)
        print(name, n, "->", len(sp))


if __name__ == "__main__":
    main()

"""The floating-point canon of the squared distance is a build-time switch on BOTH sides.

`CL3D_D2_FORM` = 0 (hipcc's contraction of masked_ordered_ball_query_gpu.cu:56-57 on gfx950, the default), 1 (no
contraction) or 2 (full fma chain, what nvcc normally emits) selects the operation order in the oracle
(oracle/cl3d_oracle.c) and in the engine (csrc/cl3d_common.h:dist2 -> libcl3d_d2form<N>.so).  CPU: the forms really
differ on near-boundary inputs (so the switch is observable).  GPU: the whole bit-exact ball-query / nearest-query
suite is re-run in a child process under forms 1 and 2, engine variant against oracle variant.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_form(form):
    import ctypes
    from oracle import native
    return ctypes.CDLL(native.build(form=form))


def _ball_query(lib, q, s, radius, K):
    import ctypes
    B, M, _ = q.shape
    N = s.shape[1]
    qm = np.ones((B, M), np.int32)
    sm = np.ones((B, N), np.int32)
    idx = np.zeros((B, M, K), np.int32)
    msk = np.zeros((B, M, K), np.int32)
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    rc = lib.oracle_masked_ordered_ball_query(q.ctypes.data_as(fp), s.ctypes.data_as(fp), qm.ctypes.data_as(ip),
                                              sm.ctypes.data_as(ip), B, M, N, ctypes.c_float(radius), K,
                                              idx.ctypes.data_as(ip), msk.ctypes.data_as(ip))
    assert rc == 0
    return idx, msk


def test_forms_are_observable_on_the_cpu():
    """Many near-equidistant neighbours: the three operation orders round differently often enough that the sorted
    index lists differ somewhere -- otherwise the switch (and the GPU test below) would prove nothing."""
    rng = np.random.default_rng(5)
    n = 4096
    d = rng.standard_normal((1, n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    s = (0.5 + 0.1 * d * (1 + 1e-6 * rng.standard_normal((1, n, 1)))).astype(np.float32)  # a thin shell around the query
    q = np.full((1, 1, 3), 0.5, np.float32)
    got = [_ball_query(_oracle_form(f), q, s, 0.2, 64)[0] for f in (0, 1, 2)]
    assert not np.array_equal(got[0], got[2]) or not np.array_equal(got[0], got[1])
    for g in got:  # all three are valid answers: 64 distinct points of the shell
        assert len(set(g[0, 0].tolist())) == 64


@pytest.mark.gpu
@pytest.mark.parametrize("form", [1, 2])
def test_bit_exact_suite_under_other_forms(form):
    from closerlook3d_amd import build
    build.build(d2_form=form)  # no-op when the variant library is newer than every source (it travels prebuilt)
    env = dict(os.environ, CL3D_D2_FORM=str(form))
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "gpu",
           os.path.join(ROOT, "tests", "test_native_gpu.py"), "-k", "ball_query or nearest_query"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout

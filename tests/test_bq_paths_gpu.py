"""The three implementations of masked_ordered_ball_query return the same bits.

`cl3d_masked_ordered_ball_query` picks by size: the single-launch search with the cell-sorted cloud resident in LDS
(csrc/ball_query_lds.hip, N and M <= 4096), the cell grid through HBM scratch (csrc/ball_query_cells.hip) or the
exhaustive scan (csrc/ball_query.hip).  `CL3D_BQ_PATH` pins one of them where it applies; the bit-exact suite of
tests/test_native_gpu.py (engine vs oracle: masked_ordered_ball_query_gpu.cu:11-96 restated in oracle/cl3d_oracle.c)
is re-run in a child process under each pin, so the cell-grid and exhaustive kernels keep their coverage on the
shapes the LDS-resident kernel now takes by default.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["cells", "exhaustive", "tile"])
def test_bit_exact_suite_under_pinned_path(path):
    env = dict(os.environ, CL3D_BQ_PATH=path)
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "gpu",
           os.path.join(ROOT, "tests", "test_native_gpu.py"), "-k", "ball_query and not stress"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout

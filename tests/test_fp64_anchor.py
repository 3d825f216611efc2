"""float64 anchor of the operator fixtures (tests/golden/make_fp64_anchor.py: the reference's own modules run in
double precision on the fixtures' inputs and parameters).

BASELINE.md section 2: "aggregated features and gradients within 1e-5 (fp32)".  Two float32 evaluations of the same
sums differ from each other by more than either differs from the exact value, so gradients are held to the bound that
means something: an implementation may be at most TWICE as far from the anchor as the reference's own float32 run is
at its worst element (+ 1e-6 absolute / relative slack for entries the reference happens to hit exactly):

    |x - anchor| <= 2 * max|reference_f32 - anchor| + 1e-6 * (1 + |anchor|)        element-wise

Here (CPU) the oracle is held to it; tests/test_fp64_anchor_gpu.py holds the engine to it.
"""
import os

import numpy as np
import pytest

from tests.helpers import GOLDEN, load_fixture, operator_fixtures, oracle_operator


def load_anchor(name):
    z = np.load(os.path.join(GOLDEN, name.replace("operators_", "fp64_anchor_")), allow_pickle=False)
    return {k: z[k] for k in z.files}


def assert_as_close_as_reference(got, ref32, anchor, what, factor=2.0, slack=1e-6):
    """got, ref32: float32 results of the implementation under test and of the reference; anchor: float64 truth."""
    got = np.asarray(got, np.float64)
    ref32 = np.asarray(ref32, np.float64)
    anchor = np.asarray(anchor, np.float64)
    assert got.shape == anchor.shape == ref32.shape, f"{what}: shapes {got.shape} {ref32.shape} {anchor.shape}"
    e_ref = float(np.abs(ref32 - anchor).max())
    err = np.abs(got - anchor)
    bound = factor * e_ref + slack * (1.0 + np.abs(anchor))
    if not (err <= bound).all():
        i = np.unravel_index(np.argmax(err - bound), err.shape)
        raise AssertionError(f"{what}: at {i} |got - anchor| = {err[i]:.3e} > {bound[i]:.3e} "
                             f"(reference's worst error {e_ref:.3e}; got {got[i]!r}, anchor {anchor[i]!r})")
    return float(err.max()), e_ref


def test_anchor_files_cover_every_fixture():
    for name in operator_fixtures() + ["operators_strided_bottleneck.npz", "operators_resnet_seg_pospool.npz",
                                       "operators_resnet_seg_pointwisemlp.npz"]:
        a = load_anchor(name)
        assert a["out64"].dtype == np.float64 and a["grad_features64"].dtype == np.float64, name


def test_reference_float32_is_within_1e5_of_the_anchor_on_single_operators():
    """The contract's 1e-5 is meaningful for one operator: the reference's float32 run itself sits within it."""
    for name in operator_fixtures():
        fx, a = load_fixture(name), load_anchor(name)
        assert np.abs(fx["out"] - a["out64"]).max() < 1e-5, name
        assert np.abs(fx["grad_features"] - a["grad_features64"]).max() < 1e-5, name


@pytest.mark.parametrize("name", operator_fixtures())
def test_oracle_is_as_close_to_the_anchor_as_the_reference(name):
    fx, a = load_fixture(name), load_anchor(name)
    out, gf, grads = oracle_operator(fx)
    assert_as_close_as_reference(out.numpy(), fx["out"], a["out64"], f"{name} out")
    assert_as_close_as_reference(gf.numpy(), fx["grad_features"], a["grad_features64"], f"{name} grad_features")
    pre = "local_aggregation_operator."
    for k, g in grads.items():
        key = k if k.startswith(pre) else pre + k
        if "grad__" + key in fx and "grad64__" + key in a:
            assert_as_close_as_reference(g.numpy().reshape(fx["grad__" + key].shape), fx["grad__" + key],
                                         a["grad64__" + key], f"{name} grad {key}")

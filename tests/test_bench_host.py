"""Host-side pieces of bench.py that need no GPU: the kernel-name map that attaches PMC counters to C-ABI entry points,
the algorithmic-byte model of the timed step, the synthetic batch."""
import numpy as np

import bench


def test_forward_product_counters_follow_the_kernel_that_ran():
    f32 = {"cl3d::pwmlp_weights_kernel(...)": {}, "cl3d::pwmlp_rows_nolds_kernel<32>(...)": {},
           "cl3d::mfma_gemm_kernel<0, 1, 1, 0, 1>(cl3d::GemmArgs)": {}}
    bf16 = {"cl3d::pwmlp_weights_kernel(...)": {}, "cl3d::mfma_gemm_kernel<1, 1, 1, 0, 0>(cl3d::GemmArgs)": {}}
    # f32: the LDS-free kernel ran, the staged kernel's launches belong to the two gradient products only
    assert "mfma_gemm_kernel" not in bench.entry_kernels("cl3d_pwmlp_point_gemm_fwd", f32)
    assert "pwmlp_rows_nolds_kernel" in bench.entry_kernels("cl3d_pwmlp_point_gemm_fwd", f32)
    # bf16 (or a shape the LDS-free kernel does not take): the staged kernel is the forward product
    assert "mfma_gemm_kernel" in bench.entry_kernels("cl3d_pwmlp_point_gemm_fwd", bf16)
    assert bench.entry_kernels("cl3d_pwmlp_point_gemm_bwd_data", f32) == ["mfma_gemm_kernel"]
    assert bench.entry_kernels("no_such_entry", f32) == []


def test_every_mapped_entry_point_is_declared_in_the_header():
    import re
    header = open(bench.os.path.join(bench.ROOT, "include", "cl3d.h")).read()
    declared = set(re.findall(r"\b(cl3d_[a-z0-9_]+)\s*\(", header))
    for entry in bench.ENTRY_KERNELS:
        assert entry in declared, entry


def test_step_byte_model_matches_survey_8d_at_the_metric_shape():
    B, N, M, K, C = 16, 4096, 4096, 32, 64
    model = bench.step_model_bytes(B, N, M, K, C)
    alg, l2, bound = model["cl3d_pwmlp_stats"]
    # one G row (C floats) per slot through the cache hierarchy
    assert l2 == B * M * K * 4 * C
    assert bound.startswith("l2-gather")
    # ght [B,N,2C] in, three [B,M,C] row tensors out (+ one byte per element of k*), idx, coordinates, masks
    assert alg == B * (4 * N * 2 * C + 4 * M * K + 16 * (M + N) + 2 * 4 * M * C + M * C)
    # the ball query reads coordinates + masks and writes idx + idx_mask
    assert model["cl3d_masked_ordered_ball_query"][0] == B * (16 * (M + N) + 8 * M * K)


def test_synthetic_batch_is_deterministic_and_shaped_like_the_metric():
    xyz, mask, feats = bench.synth_batch(2, 256, 12, 7)
    xyz2, mask2, feats2 = bench.synth_batch(2, 256, 12, 7)
    assert xyz.shape == (2, 256, 3) and mask.shape == (2, 256) and feats.shape == (2, 12, 256)
    assert xyz.dtype == np.float32 and mask.dtype == np.int32 and feats.dtype == np.float32
    assert np.array_equal(xyz, xyz2) and np.array_equal(feats, feats2) and np.array_equal(mask, mask2)

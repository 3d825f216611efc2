"""cl3d_transpose ([B,R,C] -> [B,C,R], the channel-major <-> point-major change at the fused operators' boundary)
against torch: the 16-byte path with shape-following tiles, its edge tiles, and the 4-byte fallback (extents that
are not multiples of four, misaligned views)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(16, 64, 4096), (16, 4096, 64), (2, 72, 4096), (2, 4096, 72), (3, 144, 1000), (1, 1000, 144), (2, 288, 260),
          (1, 100, 40960), (4, 36, 36), (2, 4, 8), (2, 97, 64), (2, 64, 97), (1, 3, 5), (1, 1, 7), (2, 576, 16)]


@pytest.mark.parametrize("B,R,C", SHAPES)
def test_transpose_matches_torch(B, R, C):
    from closerlook3d_amd import fused
    g = torch.Generator(device="cuda").manual_seed(R * 131 + C)
    x = torch.randn(B, R, C, device="cuda", generator=g)
    y = fused._transposed(x)
    torch.cuda.synchronize()
    assert y.shape == (B, C, R)
    assert torch.equal(y, x.transpose(1, 2).contiguous())


def test_transpose_of_a_misaligned_view_takes_the_scalar_path():
    from closerlook3d_amd import fused
    base = torch.randn(2 * 64 * 128 + 1, device="cuda")
    x = base[1:].view(2, 64, 128)  # 4-byte aligned only
    y = fused._transposed(x)
    torch.cuda.synchronize()
    assert torch.equal(y, x.transpose(1, 2).contiguous())

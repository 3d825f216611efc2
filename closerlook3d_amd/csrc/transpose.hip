// transpose.hip -- [B,R,C] -> [B,C,R] float32, the layout change at the boundary of the fused operators
// (the reference's tensors are channel-major [B,C,N]; the fused kernels gather point-major rows [B,N,C]).
// Tiles of TR x TC elements through LDS (row stride TC + 1: conflict-free both ways).  When both extents are
// multiples of four (and the pointers 16-byte aligned) every global access is a 16-byte vector -- four consecutive
// columns of a source row on the way in, four consecutive source rows of one column on the way out -- and the tile
// extents follow the shape (72 channels are ONE tile of 72 rows, not 64 + 8; tile_shape): round 2's kernel moved 4 bytes per
// lane in fixed 64 x 64 tiles and reached 1.1 TB/s on the [16,64,4096] tensors of the operator benches (30 us for
// 33.5 MB; the 72-channel case ran half its workgroups on 8-row tiles).
#include "cl3d_common.h"

namespace cl3d {

constexpr int kTrMax = 96;         // largest tile extent
constexpr int kTrFloats = 3200;    // largest padded tile (12.5 KB of LDS: see tile_shape)

// PRO: element (r, c) enters as max(row_scale[r] * x + row_shift[r], 0) -- the BatchNorm + ReLU of the layer that produced a
// channel-major tensor (r = channel), applied in the layout change instead of in a pass of its own (round 5)
template <bool PRO>
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        int R, int C, const float *__restrict__ row_scale,
                                                        const float *__restrict__ row_shift) {
  __shared__ float tile[64 * 65];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float *s = src + (size_t)b * R * C;
  float *d = dst + (size_t)b * R * C;
  float v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {  // all 16 loads of the thread in flight
    const int r = r0 + ty + 4 * u, c = c0 + tx;
    v[u] = s[(size_t)(r < R ? r : R - 1) * C + (c < C ? c : C - 1)];
    if (PRO) {
      const float z = __builtin_fmaf(v[u], row_scale[r < R ? r : R - 1], row_shift[r < R ? r : R - 1]);
      v[u] = z > 0.f ? z : 0.f;
    }
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) tile[(ty + 4 * u) * 65 + tx] = v[u];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int c = c0 + ty + 4 * u, r = r0 + tx;
    if (c < C && r < R) d[(size_t)c * R + r] = tile[tx * 65 + ty + 4 * u];
  }
}

// R % 4 == 0, C % 4 == 0, TR % 4 == 0, TC % 4 == 0, TR, TC <= kTrMax
template <bool PRO>
__global__ __launch_bounds__(256) void transpose4_kernel(const float *__restrict__ src, float *__restrict__ dst, int R,
                                                         int C, int TR, int TC, const float *__restrict__ row_scale,
                                                         const float *__restrict__ row_shift) {
  __shared__ float tile[kTrFloats];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * TR, c0 = blockIdx.x * TC;
  const float *s = src + (size_t)b * R * C;
  float *d = dst + (size_t)b * R * C;
  const int ld = TC + 1;
  const int cq = TC / 4, rq = TR / 4;
  constexpr int kMaxV = (kTrFloats / 4 + 255) / 256;  // float4 per thread, at most
  float4 v[kMaxV];
#pragma unroll
  for (int u = 0; u < kMaxV; ++u) {  // every load of the thread in flight
    const int e = u * 256 + (int)threadIdx.x;
    const int r = e / cq, c4 = e - r * cq;
    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < TR * cq && r0 + r < R && c0 + 4 * c4 < C) {
      v[u] = *reinterpret_cast<const float4 *>(s + (size_t)(r0 + r) * C + c0 + 4 * c4);
      if (PRO) {
        const float sc = row_scale[r0 + r], sh = row_shift[r0 + r];
        v[u].x = __builtin_fmaf(v[u].x, sc, sh); v[u].y = __builtin_fmaf(v[u].y, sc, sh);
        v[u].z = __builtin_fmaf(v[u].z, sc, sh); v[u].w = __builtin_fmaf(v[u].w, sc, sh);
        v[u].x = v[u].x > 0.f ? v[u].x : 0.f; v[u].y = v[u].y > 0.f ? v[u].y : 0.f;
        v[u].z = v[u].z > 0.f ? v[u].z : 0.f; v[u].w = v[u].w > 0.f ? v[u].w : 0.f;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < kMaxV; ++u) {
    const int e = u * 256 + (int)threadIdx.x;
    const int r = e / cq, c4 = e - r * cq;
    if (e < TR * cq) {
      float *t = tile + r * ld + 4 * c4;
      t[0] = v[u].x; t[1] = v[u].y; t[2] = v[u].z; t[3] = v[u].w;
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kMaxV; ++u) {
    const int e = u * 256 + (int)threadIdx.x;
    const int c = e / rq, r4 = e - c * rq;
    if (e < TC * rq && c0 + c < C && r0 + 4 * r4 < R) {
      const float *t = tile + (4 * r4) * ld + c;
      *reinterpret_cast<float4 *>(d + (size_t)(c0 + c) * R + r0 + 4 * r4) = make_float4(t[0], t[ld], t[2 * ld], t[3 * ld]);
    }
  }
}

// tile extent for an axis of n elements: the whole axis if it fits, else an even split into pieces <= 64 + slack
static int tile_extent(int n) {
  if (n <= kTrMax) return (n + 3) & ~3;
  const int pieces = (n + 63) / 64;
  int t = (n + pieces - 1) / pieces;
  t = (t + 3) & ~3;
  return t > kTrMax ? 64 : t;
}

// Tile of a [R,C] plane.  The padded tile stays under 12.5 KB: the layout changes of an operator's forward pass are
// queued beside the ball query, whose workgroup leaves ~13 KB of a CU's LDS free (csrc/ball_query_lds.hip) -- a 37 KB
// tile (96 x 97, rounds 2-3) only got onto a CU when a ball-query workgroup retired (replayed PosPool step: the
// transpose of the features ended 4 us AFTER the ball query instead of 27 us before its end; the gather pass that
// waits for both starts one cross-queue hop after the ball query either way, so the step gained nothing measurable
// there, PseudoGrid 1.2 %).  The long axis gives way first
// (32 elements = one 128-byte line per row of the tile), then the short one.
static void tile_shape(int R, int C, int *tr, int *tc) {
  int TR = tile_extent(R), TC = tile_extent(C);
  const auto fits = [](int a, int b) { return a * (b + 1) <= kTrFloats; };
  if (!fits(TR, TC) && C > kTrMax) TC = 32;
  if (!fits(TR, TC) && R > kTrMax) TR = 32;
  while (!fits(TR, TC)) {  // both axes short and the plane still too large: halve the longer extent
    if (TR >= TC) TR = ((TR / 2) + 3) & ~3; else TC = ((TC / 2) + 3) & ~3;
  }
  *tr = TR;
  *tc = TC;
}

}  // namespace cl3d

static int transpose_launch(const float *src, int B, int R, int C, float *dst, const float *row_scale,
                            const float *row_shift, hipStream_t st, const char *who) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 0 && R >= 0 && C >= 0, "transpose: bad sizes");
  if (B == 0 || R == 0 || C == 0) return CL3D_OK;
  CL3D_REQUIRE(src && dst, "transpose: null pointer");
  CL3D_REQUIRE(B <= 65535, "transpose: grid limit");
  const bool pro = row_scale != nullptr;
  const bool vec = (R & 3) == 0 && (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
  if (vec) {
    int TR, TC;
    tile_shape(R, C, &TR, &TC);
    CL3D_REQUIRE(ceil_div(R, TR) <= 65535, "transpose: grid limit");
    const dim3 grid(ceil_div(C, TC), ceil_div(R, TR), B);
    if (pro) hipLaunchKernelGGL((transpose4_kernel<true>), grid, dim3(256), 0, st, src, dst, R, C, TR, TC, row_scale, row_shift);
    else hipLaunchKernelGGL((transpose4_kernel<false>), grid, dim3(256), 0, st, src, dst, R, C, TR, TC, row_scale, row_shift);
    return check_launch(who);
  }
  CL3D_REQUIRE(ceil_div(R, 64) <= 65535, "transpose: grid limit");
  const dim3 grid(ceil_div(C, 64), ceil_div(R, 64), B);
  if (pro) hipLaunchKernelGGL((transpose_kernel<true>), grid, dim3(256), 0, st, src, dst, R, C, row_scale, row_shift);
  else hipLaunchKernelGGL((transpose_kernel<false>), grid, dim3(256), 0, st, src, dst, R, C, row_scale, row_shift);
  return check_launch(who);
}

extern "C" int cl3d_transpose(const float *src, int B, int R, int C, float *dst, cl3d_stream_t stream) {
  return transpose_launch(src, B, R, C, dst, nullptr, nullptr, (hipStream_t)stream, "cl3d_transpose");
}

// [B,R,C] -> [B,C,R] with max(row_scale[r] * x + row_shift[r], 0) applied on the way: a channel-major tensor's folded
// BatchNorm + ReLU (r = channel) in the layout change that turns it into point-major rows
extern "C" int cl3d_transpose_bn_relu(const float *src, const float *row_scale, const float *row_shift, int B, int R, int C,
                                      float *dst, cl3d_stream_t stream) {
  CL3D_REQUIRE(row_scale && row_shift, "transpose_bn_relu: null scale / shift");
  return transpose_launch(src, B, R, C, dst, row_scale, row_shift, (hipStream_t)stream, "cl3d_transpose_bn_relu");
}

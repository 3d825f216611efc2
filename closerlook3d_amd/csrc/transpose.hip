// transpose.hip -- [B,R,C] -> [B,C,R] float32, the layout change at the boundary of the fused operators
// (the reference's tensors are channel-major [B,C,N]; the fused kernels gather point-major rows [B,N,C]).
// 64x64 tiles through LDS (row stride 65: conflict-free both ways), 256-byte row segments on both sides.
#include "cl3d_common.h"

namespace cl3d {

__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                        int R, int C) {
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

}  // namespace cl3d

extern "C" int cl3d_transpose(const float *src, int B, int R, int C, float *dst, cl3d_stream_t stream) {
  CL3D_REQUIRE(B >= 0 && R >= 0 && C >= 0, "transpose: bad sizes");
  if (B == 0 || R == 0 || C == 0) return CL3D_OK;
  CL3D_REQUIRE(src && dst, "transpose: null pointer");
  CL3D_REQUIRE(B <= 65535 && cl3d::ceil_div(R, 64) <= 65535, "transpose: grid limit");
  hipLaunchKernelGGL(cl3d::transpose_kernel, dim3(cl3d::ceil_div(C, 64), cl3d::ceil_div(R, 64), B), dim3(256), 0,
                     (hipStream_t)stream, src, dst, R, C);
  return cl3d::check_launch("cl3d_transpose");
}

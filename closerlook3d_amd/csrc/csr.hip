// csr.hip -- CSR inverse of a neighbour-index tensor: for every support point, the ascending list of
// (query, neighbour-slot) positions that reference it.
//
// idx [B, MK] (MK = M*K flattened slots) with values in [0,N)  ->
//   inv_off   [B, N+1]   segment starts (inv_off[b][N] == MK)
//   inv_slots [B, MK]    slot ids, ascending inside each segment
//
// This is what turns every backward scatter of the fused operators into an ordered gather (no float
// atomics, summation order fixed).  It depends on idx only, so one build serves the backward of every
// operator that shares the ball query.  Steps: integer histogram (atomics on ints are exact, order
// does not matter) -> per-cloud exclusive scan -> atomic-cursor fill (unordered inside a segment) ->
// per-segment wave rank sort (restores ascending slot order => deterministic).
#include "cl3d_common.h"

namespace cl3d {

__global__ __launch_bounds__(256) void csr_count_kernel(const int *__restrict__ idx, int N, int MK,
                                                        int *__restrict__ cnt) {
  const int b = blockIdx.y;
  const int *ib = idx + (size_t)b * MK;
  int *cb = cnt + (size_t)b * N;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < MK; e += gridDim.x * 256) {
    const int i = ib[e];
    if ((unsigned)i < (unsigned)N) atomicAdd(&cb[i], 1);
  }
}

// one block per cloud: off = exclusive scan of cnt; cnt is overwritten with the same values (fill cursors)
__global__ __launch_bounds__(1024) void csr_scan_kernel(int *__restrict__ cnt, int N,
                                                        int *__restrict__ off) {
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  const int b = blockIdx.x;
  int *cb = cnt + (size_t)b * N;
  int *ob = off + (size_t)b * (N + 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < N ? cb[i] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_wave[w];
    const int carry = s_carry;
    const int excl = carry + woff + incl - v;
    if (i < N) {
      ob[i] = excl;
      cb[i] = excl;
    }
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = excl + v;
    __syncthreads();
  }
  if (threadIdx.x == 0) ob[N] = s_carry;
}

__global__ __launch_bounds__(256) void csr_fill_kernel(const int *__restrict__ idx, int N, int MK,
                                                       int *__restrict__ cursor,
                                                       int *__restrict__ tmp) {
  const int b = blockIdx.y;
  const int *ib = idx + (size_t)b * MK;
  int *cb = cursor + (size_t)b * N;
  int *tb = tmp + (size_t)b * MK;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < MK; e += gridDim.x * 256) {
    const int i = ib[e];
    if ((unsigned)i < (unsigned)N) tb[atomicAdd(&cb[i], 1)] = e;
  }
}

// one wave per segment: rank sort (values are unique slot ids)
__global__ __launch_bounds__(256) void csr_sort_kernel(const int *__restrict__ off,
                                                       const int *__restrict__ tmp, int N, int MK,
                                                       int *__restrict__ slots) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const int *ob = off + (size_t)b * (N + 1);
  const int s0 = ob[i], s1 = ob[i + 1];
  const int len = s1 - s0;
  const int *tb = tmp + (size_t)b * MK + s0;
  int *sb = slots + (size_t)b * MK + s0;
  if (len <= 64) {
    const int v = lane < len ? tb[lane] : 0x7fffffff;
    int rank = 0;
    for (int t = 0; t < len; ++t) rank += (__shfl(v, t, 64) < v) ? 1 : 0;
    if (lane < len) sb[rank] = v;
  } else {
    for (int e = lane; e < len; e += 64) {
      const int v = tb[e];
      int rank = 0;
      for (int t = 0; t < len; ++t) rank += (tb[t] < v) ? 1 : 0;
      sb[rank] = v;
    }
  }
}

}  // namespace cl3d

extern "C" int cl3d_build_inverse_index(const int32_t *idx, int B, int N, int MK, int32_t *inv_off,
                                        int32_t *inv_slots, void *ws, size_t ws_bytes,
                                        cl3d_stream_t stream) {
  CL3D_REQUIRE(B >= 0 && N >= 1 && MK >= 0, "build_inverse_index: bad sizes");
  if (B == 0) return CL3D_OK;
  CL3D_REQUIRE(idx || MK == 0, "build_inverse_index: null idx");
  CL3D_REQUIRE(inv_off && (inv_slots || MK == 0), "build_inverse_index: null output");
  CL3D_REQUIRE(B <= 65535, "build_inverse_index: B exceeds grid.y limit");
  const size_t need = ((size_t)B * N + (size_t)B * MK) * sizeof(int);
  if (ws_bytes < need || !ws) return cl3d::fail(CL3D_E_WORKSPACE, "build_inverse_index: workspace %zu < %zu", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  int *cnt = static_cast<int *>(ws);
  int *tmp = cnt + (size_t)B * N;
  hipError_t e = hipMemsetAsync(cnt, 0, (size_t)B * N * sizeof(int), st);
  if (e != hipSuccess) return cl3d::fail(CL3D_E_LAUNCH, "build_inverse_index: memset: %s", hipGetErrorString(e));
  int gx = cl3d::ceil_div(MK > 0 ? MK : 1, 256 * 4);
  gx = gx > 2048 ? 2048 : gx;
  if (MK > 0) hipLaunchKernelGGL(cl3d::csr_count_kernel, dim3(gx, B), dim3(256), 0, st, idx, N, MK, cnt);
  hipLaunchKernelGGL(cl3d::csr_scan_kernel, dim3(B), dim3(1024), 0, st, cnt, N, inv_off);
  if (MK > 0) {
    hipLaunchKernelGGL(cl3d::csr_fill_kernel, dim3(gx, B), dim3(256), 0, st, idx, N, MK, cnt, tmp);
    hipLaunchKernelGGL(cl3d::csr_sort_kernel, dim3(cl3d::ceil_div(N, 4), B), dim3(256), 0, st, inv_off, tmp, N, MK, inv_slots);
  }
  return cl3d::check_launch("cl3d_build_inverse_index");
}

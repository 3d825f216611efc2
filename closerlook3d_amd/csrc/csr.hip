// csr.hip -- CSR inverse of a neighbour-index tensor: for every support point, the ascending list of
// (query, neighbour-slot) positions that reference it.
//
// idx [B, MK] (MK = M*K flattened slots) with values in [0,N)  ->
//   inv_off   [B, N+1]   segment starts (inv_off[b][N] = number of valid slots of cloud b)
//   inv_slots [B, MK]    slot ids, ascending inside each segment (invalid indices sort to the tail)
//
// This is what turns every backward scatter of the fused operators into an ordered gather (no float
// atomics, summation order fixed).  It depends on idx only, so one build serves the backward of every
// operator that shares the ball query.
//
// Build = one stable LSD radix sort of (key = cloud*(N+1) + idx, value = slot) pairs + a binary search
// per row for the segment starts.  The sort is rocPRIM's device radix sort (a generic primitive, like
// the library GEMM); stability gives ascending slot ids inside a segment, hence a deterministic order.
// History (metric shape, 2.1 M slots, per build): global integer atomics + per-segment rank sort
// 270 us; row-ownership scans (every wave streams the whole slot array for its 32-64 rows, from L2 or
// through LDS, ordered or cursor-based fill) 230-350 us -- the work is O(MK * N/rows) compare
// instructions however it is staged.  A radix sort is O(MK).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "cl3d_common.h"

namespace cl3d {

__global__ __launch_bounds__(256) void csr_keys_kernel(const int *__restrict__ idx, int N, int MK, long long total,
                                                       unsigned *__restrict__ keys, int *__restrict__ vals) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long long)gridDim.x * 256) {
    const int b = (int)(p / MK);
    const int e = (int)(p - (long long)b * MK);
    const int i = idx[p];
    keys[p] = (unsigned)b * (unsigned)(N + 1) + ((unsigned)i < (unsigned)N ? (unsigned)i : (unsigned)N);
    vals[p] = e;
  }
}

// off[b][i] = (first position in cloud b's sorted keys with key >= b*(N+1)+i) - b*MK
__global__ __launch_bounds__(256) void csr_offsets_kernel(const unsigned *__restrict__ sorted_keys, int B, int N,
                                                          int MK, int *__restrict__ off) {
  const long long rows = (long long)B * (N + 1);
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
    const int b = (int)(r / (N + 1));
    const unsigned key = (unsigned)r;  // == b*(N+1) + i
    const unsigned *k = sorted_keys + (size_t)b * MK;
    int lo = 0, hi = MK;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (k[mid] < key) lo = mid + 1;
      else hi = mid;
    }
    off[r] = lo;
  }
}

static unsigned key_bits(int B, int N) {
  const unsigned long long maxkey = (unsigned long long)B * (unsigned long long)(N + 1);
  unsigned bits = 1;
  while ((1ull << bits) < maxkey && bits < 32) ++bits;
  return bits;
}

static size_t sort_temp_bytes(int B, int N, int MK) {
  size_t bytes = 0;
  const unsigned n = (unsigned)((size_t)B * MK);
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned *)nullptr, (unsigned *)nullptr,
                                  (const int *)nullptr, (int *)nullptr, n, 0, key_bits(B, N), (hipStream_t)0);
  return bytes;
}

size_t inverse_index_workspace(int B, int N, int MK) {
  const size_t n = (size_t)B * MK;
  if (n == 0) return 0;
  return 3 * ((n * 4 + 255) & ~(size_t)255) + sort_temp_bytes(B, N, MK);
}

}  // namespace cl3d

extern "C" int cl3d_build_inverse_index(const int32_t *idx, int B, int N, int MK, int32_t *inv_off,
                                        int32_t *inv_slots, void *ws, size_t ws_bytes,
                                        cl3d_stream_t stream) {
  CL3D_REQUIRE(B >= 0 && N >= 1 && MK >= 0, "build_inverse_index: bad sizes");
  if (B == 0) return CL3D_OK;
  CL3D_REQUIRE(idx || MK == 0, "build_inverse_index: null idx");
  CL3D_REQUIRE(inv_off && (inv_slots || MK == 0), "build_inverse_index: null output");
  CL3D_REQUIRE((unsigned long long)B * (unsigned long long)(N + 1) <= 0xffffffffull && (size_t)B * MK <= 0x7fffffffu,
               "build_inverse_index: problem too large for 32-bit keys");
  hipStream_t st = (hipStream_t)stream;
  if (MK == 0) {
    hipError_t e = hipMemsetAsync(inv_off, 0, (size_t)B * (N + 1) * sizeof(int), st);
    return e == hipSuccess ? CL3D_OK : cl3d::fail(CL3D_E_LAUNCH, "build_inverse_index: memset: %s", hipGetErrorString(e));
  }
  const size_t need = cl3d::inverse_index_workspace(B, N, MK);
  if (ws_bytes < need || !ws) return cl3d::fail(CL3D_E_WORKSPACE, "build_inverse_index: workspace %zu < %zu", ws_bytes, need);
  const size_t n = (size_t)B * MK;
  const size_t stride = (n * 4 + 255) & ~(size_t)255;
  char *p = static_cast<char *>(ws);
  unsigned *keys_in = reinterpret_cast<unsigned *>(p);
  unsigned *keys_out = reinterpret_cast<unsigned *>(p + stride);
  int *vals_in = reinterpret_cast<int *>(p + 2 * stride);
  void *temp = p + 3 * stride;
  size_t temp_bytes = ws_bytes - 3 * stride;
  int gx = (int)((n + 256 * 8 - 1) / (256 * 8));
  gx = gx > 4096 ? 4096 : (gx < 1 ? 1 : gx);
  hipLaunchKernelGGL(cl3d::csr_keys_kernel, dim3(gx), dim3(256), 0, st, idx, N, MK, (long long)n, keys_in, vals_in);
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, (const unsigned *)keys_in, keys_out, (const int *)vals_in,
                                           inv_slots, (unsigned)n, 0, cl3d::key_bits(B, N), st);
  if (e != hipSuccess) return cl3d::fail(CL3D_E_LAUNCH, "build_inverse_index: radix sort: %s", hipGetErrorString(e));
  const long long rows = (long long)B * (N + 1);
  int gr = (int)((rows + 255) / 256);
  gr = gr > 4096 ? 4096 : gr;
  hipLaunchKernelGGL(cl3d::csr_offsets_kernel, dim3(gr), dim3(256), 0, st, keys_out, B, N, MK, inv_off);
  return cl3d::check_launch("cl3d_build_inverse_index");
}

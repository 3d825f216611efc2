// csr.hip -- CSR inverse of a neighbour-index tensor: for every support point, the ascending list of
// (query, neighbour-slot) positions that reference it.
//
// idx [B, MK] (MK = M*K flattened slots) with values in [0,N)  ->
//   inv_off   [B, N+1]   segment starts (inv_off[b][N] == MK)
//   inv_slots [B, MK]    slot ids, ascending inside each segment
//
// This is what turns every backward scatter of the fused operators into an ordered gather (no float
// atomics, summation order fixed).  It depends on idx only, so one build serves the backward of every
// operator that shares the ball query.
//
// Build = two scans of idx + one prefix sum, no global atomics and no sort:
//   count  every wave owns 32 consecutive support rows and streams the cloud's whole slot array (0.5 MB
//          at the metric shape, L2-resident and shared by all waves of the cloud), counting the slots
//          that land in its rows (LDS integer atomics: exact, 8 op/clk/CU);
//   scan   per-cloud exclusive prefix sum of the counts;
//   fill   the same stream again; hits are handled one at a time in slot order (wave-uniform loop over
//          the ballot), so each row's list comes out ascending by construction.
// (A first version used global integer atomics + a per-segment rank sort: 270 us per build at the
//  metric shape, against ~80 us for this one.)
#include "fused_common.h"

namespace cl3d {

constexpr int kCsrRows = 32;  // support rows per wave
constexpr int kCsrUnroll = 8;

__global__ __launch_bounds__(256) void csr_count_kernel(const int *__restrict__ idx, int B, int N, int MK,
                                                        int *__restrict__ cnt) {
  __shared__ unsigned s_cnt[4][kCsrRows];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int tiles_per_cloud = (N + 4 * kCsrRows - 1) / (4 * kCsrRows);
  int b, tile;
  decode_tile(blockIdx.x, B, tiles_per_cloud, b, tile);
  const int r0 = (tile * 4 + wave) * kCsrRows;
  if (lane < kCsrRows) s_cnt[wave][lane] = 0u;
  __syncthreads();
  if (r0 < N) {
    const int *ib = idx + (size_t)b * MK;
    // kCsrUnroll independent loads in flight per lane: the loop is otherwise bound by one L2 round trip
    // per 64 slots
    int base = 0;
    for (; base + kCsrUnroll * 64 <= MK; base += kCsrUnroll * 64) {
      int d[kCsrUnroll];
#pragma unroll
      for (int u = 0; u < kCsrUnroll; ++u) d[u] = ib[base + u * 64 + lane] - r0;
#pragma unroll
      for (int u = 0; u < kCsrUnroll; ++u)
        if ((unsigned)d[u] < (unsigned)kCsrRows) atomicAdd(&s_cnt[wave][d[u]], 1u);
    }
    for (; base < MK; base += 64) {
      const int e = base + lane;
      const int d = (e < MK ? ib[e] : -1) - r0;
      if ((unsigned)d < (unsigned)kCsrRows) atomicAdd(&s_cnt[wave][d], 1u);
    }
  }
  __syncthreads();
  if (lane < kCsrRows && r0 + lane < N) cnt[(size_t)b * N + r0 + lane] = (int)s_cnt[wave][lane];
}

// one block per cloud: off = exclusive scan of cnt
__global__ __launch_bounds__(1024) void csr_scan_kernel(const int *__restrict__ cnt, int N,
                                                        int *__restrict__ off) {
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  const int b = blockIdx.x;
  const int *cb = cnt + (size_t)b * N;
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
    const int excl = s_carry + woff + incl - v;
    if (i < N) ob[i] = excl;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = excl + v;
    __syncthreads();
  }
  if (threadIdx.x == 0) ob[N] = s_carry;
}

__global__ __launch_bounds__(256) void csr_fill_kernel(const int *__restrict__ idx, const int *__restrict__ off,
                                                       int B, int N, int MK, int *__restrict__ slots) {
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int tiles_per_cloud = (N + 4 * kCsrRows - 1) / (4 * kCsrRows);
  int b, tile;
  decode_tile(blockIdx.x, B, tiles_per_cloud, b, tile);
  const int r0 = (tile * 4 + wave) * kCsrRows;
  if (r0 >= N) return;
  const int *ib = idx + (size_t)b * MK;
  int *sb = slots + (size_t)b * MK;
  // lane r (< 32) carries the next write position of row r0 + r
  int pos = (lane < kCsrRows && r0 + lane < N) ? off[(size_t)b * (N + 1) + r0 + lane] : 0;
  auto emit = [&](int d, int base) {
    unsigned long long m = __ballot((unsigned)d < (unsigned)kCsrRows);
    while (m) {  // wave-uniform: hits in ascending slot order
      const int t = __builtin_ctzll(m);
      m &= m - 1;
      const int row = __builtin_amdgcn_readlane(d, t);
      const int p = __builtin_amdgcn_readlane(pos, row);
      if (lane == row) pos += 1;
      if (lane == 0) sb[p] = base + t;
    }
  };
  int base = 0;
  for (; base + kCsrUnroll * 64 <= MK; base += kCsrUnroll * 64) {
    int d[kCsrUnroll];
#pragma unroll
    for (int u = 0; u < kCsrUnroll; ++u) d[u] = ib[base + u * 64 + lane] - r0;
#pragma unroll
    for (int u = 0; u < kCsrUnroll; ++u) emit(d[u], base + u * 64);
  }
  for (; base < MK; base += 64) {
    const int e = base + lane;
    emit((e < MK ? ib[e] : -1) - r0, base);
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
  const size_t need = (size_t)B * N * sizeof(int);
  if (ws_bytes < need || !ws) return cl3d::fail(CL3D_E_WORKSPACE, "build_inverse_index: workspace %zu < %zu", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  int *cnt = static_cast<int *>(ws);
  const int tiles_per_cloud = cl3d::ceil_div(N, 4 * cl3d::kCsrRows);
  const long long blocks = (long long)B * tiles_per_cloud;
  CL3D_REQUIRE(blocks <= 0x7fffffffLL, "build_inverse_index: too many rows");
  hipLaunchKernelGGL(cl3d::csr_count_kernel, dim3((unsigned)blocks), dim3(256), 0, st, idx, B, N, MK, cnt);
  hipLaunchKernelGGL(cl3d::csr_scan_kernel, dim3(B), dim3(1024), 0, st, cnt, N, inv_off);
  if (MK > 0)
    hipLaunchKernelGGL(cl3d::csr_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, idx, inv_off, B, N, MK, inv_slots);
  return cl3d::check_launch("cl3d_build_inverse_index");
}

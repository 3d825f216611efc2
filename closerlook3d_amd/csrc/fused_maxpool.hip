// fused_maxpool.hip -- neighbourhood max-pooling without the [B,C,M,K] tensor (gfx950).
//
// MaskedMaxPool (reference pt_utils.py:179-202) gathers the neighbourhood features [B,C,npoint,K]
// (up to 600 MB per call in a ModelNet backbone step) and runs F.max_pool2d over K.  Here one kernel reads
// each neighbour's point-major feature row once and keeps the running maximum in registers; the arg-max
// (first maximum, as max_pool2d's backward routes it) is kept as one byte per (query, channel).  The
// backward is an ordered gather through the CSR inverse of idx: a slot passes its query's gradient on
// exactly for the channels whose arg-max it is.  No atomics, fixed summation order.
//
// Round 6: the TARGET form.  The forward pass can leave the arg-max's SUPPORT INDEX per (channel, query) instead of its slot
// (cl3d_maxpool_fwd_targets: int32 [B,C,M], channel-major like the output), and the backward is then a scatter with exactly
// one target per (query, channel) -- cl3d_maxpool_bwd_targets = the PointWiseMLP's arg-max scatter (pwmlp_hit_kernel:
// ds_add_f64 into LDS rows of doubles, a few floats per row: exact, order-free), on the channel-major gradient as autograd
// hands it over: no CSR inverse of idx (four launches per pooling layer), no transposed copy of the gradient, M C adds
// instead of M K C tests.  85 -> ~20 us for the first pooling layer of the config-2 backbone.
#include "fused_common.h"

namespace cl3d {

// The pooling kernels keep rounds 1-5's lane maps (most rows per wave): measured against the wide maps of fused_common.h on
// the replayed backbones, config 4 5.66 against 5.75 ms, config 5 19.75 against 19.87, configs 2 / 3 the same
// (profiles/r06/session20c_summary.txt).
constexpr bool kMaxpoolWide = false;


struct MaxArgs {
  const int *idx;             // [B,M,K]
  const float *ft;            // fwd: [B,N,C] point-major features
  float *out;                 // fwd: [B,C,M] channel-major (the API layout), written directly
  unsigned char *kstar_t;     // [B,M,C]
  int *target_cm;             // fwd, nullable: [B,C,M] support index of the arg-max (the target form)
  const float *gout_t;        // bwd: [B,M,C]
  const int *inv_off, *inv_slots;
  float *dft;                 // bwd: [B,N,C], or [B,C,N] when dft_channel_major
  int dft_channel_major;
  int B, N, M, K, C;
  int L, QW, chunks;
};

template <int V>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(MaxArgs a) {
  extern __shared__ int sidx[];  // [TQ][K+1]: odd row stride keeps the lane groups of a wave on different LDS banks
  const int K = a.K, C = a.C, M = a.M, N = a.N, L = a.L, QW = a.QW;
  const int TQ = 4 * QW;
  const int KS = K + 1;
  int b, tq;
  decode_tile(blockIdx.x, a.B, (M + TQ - 1) / TQ, b, tq);
  const int j0 = tq * TQ;
  for (int t = threadIdx.x; t < TQ * K; t += 256) {
    const int j = j0 + t / K;
    sidx[(t / K) * KS + (t - (t / K) * K)] = j < M ? a.idx[((size_t)b * M + j) * K + (t - (t / K) * K)] : 0;
  }
  __syncthreads();
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int g = lane / L, cl = lane - g * L;
  if (g >= QW) return;
  const int jq = wave * QW + g;
  const int j = j0 + jq;
  if (j >= M) return;
  const int *my = sidx + jq * KS;
  const float *rows = a.ft + (size_t)b * N * C;
  constexpr int KB = 8;
  for (int ch = blockIdx.y; ch < a.chunks; ch += gridDim.y) {
    const int c0 = (ch * L + cl) * V;
    if (c0 >= C) continue;
    float best[V];
    int kb[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      best[v] = 0.f;
      kb[v] = 0;
    }
    for (int k0 = 0; k0 < K; k0 += KB) {
      Vec<V> r[KB];
#pragma unroll
      for (int u = 0; u < KB; ++u) r[u] = load_row<V>(rows + (size_t)my[k0 + u < K ? k0 + u : K - 1] * C + c0);
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int k = k0 + u;
        if (k >= K) continue;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          if (k == 0 || r[u].v[v] > best[v]) {
            best[v] = r[u].v[v];
            kb[v] = k;
          }
        }
      }
    }
    const size_t orow = ((size_t)b * M + j) * C + c0;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      a.out[((size_t)b * C + c0 + v) * M + j] = best[v];
      if (a.kstar_t) a.kstar_t[orow + v] = (unsigned char)kb[v];
      if (a.target_cm) a.target_cm[((size_t)b * C + c0 + v) * M + j] = my[kb[v]];
    }
  }
}

template <int V>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(MaxArgs a) {
  const int K = a.K, C = a.C, M = a.M, N = a.N, L = a.L, QW = a.QW;
  const int MK = M * K;
  const int TR = 4 * QW;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int g = lane / L, cl = lane - g * L;
  if (g >= QW) return;
  const int tiles_per_cloud = (N + TR - 1) / TR;
  const int ntiles = a.B * tiles_per_cloud;
  constexpr int SB = 4;
  for (int ch = blockIdx.y; ch < a.chunks; ch += gridDim.y) {
    const int c0 = (ch * L + cl) * V;
    if (c0 >= C) continue;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      int b, tr;
      decode_tile(tile, a.B, tiles_per_cloud, b, tr);
      const int i = tr * TR + wave * QW + g;
      if (i >= N) continue;
      const int *off = a.inv_off + (size_t)b * (N + 1);
      const int *slots = a.inv_slots + (size_t)b * MK;
      const float *grow = a.gout_t + (size_t)b * M * C + c0;
      const unsigned char *ksrow = a.kstar_t + (size_t)b * M * C + c0;
      const int s0 = off[i], s1 = off[i + 1];
      float acc[V];
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = 0.f;
      for (int e = s0; e < s1; e += SB) {
        int sl[SB];
        Vec<V> gg[SB];
        unsigned ksw[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) sl[u] = slots[e + u < s1 ? e + u : s1 - 1];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int j = sl[u] / K;
          gg[u] = load_row<V>(grow + (size_t)j * C);
          if constexpr (V == 4) ksw[u] = *reinterpret_cast<const unsigned *>(ksrow + (size_t)j * C);
          else ksw[u] = ksrow[(size_t)j * C];
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          if (e + u >= s1) continue;
          const int k = sl[u] - (sl[u] / K) * K;
#pragma unroll
          for (int v = 0; v < V; ++v)
            if (k == (int)((ksw[u] >> (8 * v)) & 0xffu)) acc[v] += gg[u].v[v];
        }
      }
      Vec<V> o;
#pragma unroll
      for (int v = 0; v < V; ++v) o.v[v] = acc[v];
      if (a.dft_channel_major) {
#pragma unroll
        for (int v = 0; v < V; ++v) a.dft[((size_t)b * C + c0 + v) * N + i] = o.v[v];
      } else {
        store_row<V>(a.dft + ((size_t)b * N + i) * C + c0, o);
      }
    }
  }
}

bool maxpool_supported(int K, int C) {
  if (K < 1 || K > 255 || C < 1) return false;
  LaneMap m = pick_lane_map(C, (C % 4 == 0) ? 4 : 1, 64, kMaxpoolWide);
  while (4 * (size_t)m.QW * (K + 1) * sizeof(int) > 48 * 1024 && m.QW > 1) m.QW -= 1;
  return 4 * (size_t)m.QW * (K + 1) * sizeof(int) <= 64 * 1024;
}

}  // namespace cl3d

static int maxpool_forward(const int32_t *idx, const float *ft, int B, int N, int M, int K, int C, float *out,
                           unsigned char *kstar_t, int32_t *target_cm, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 0 && N >= 1 && M >= 0 && K >= 1 && C >= 1, "maxpool_fwd: bad sizes");
  if (K > 255) return fail(CL3D_E_UNSUPPORTED, "maxpool_fwd: nsample=%d > 255 (arg-max is stored in a byte)", K);
  if (B == 0 || M == 0) return CL3D_OK;
  CL3D_REQUIRE(idx && ft && out, "maxpool_fwd: null pointer");
  MaxArgs a{};
  a.idx = idx; a.ft = ft; a.out = out; a.kstar_t = kstar_t; a.target_cm = target_cm; a.B = B; a.N = N; a.M = M; a.K = K; a.C = C;
  const int V = (C % 4 == 0) ? 4 : 1;
  LaneMap m = pick_lane_map(C, V, 64, kMaxpoolWide);
  while (4 * (size_t)m.QW * (K + 1) * sizeof(int) > 48 * 1024 && m.QW > 1) m.QW -= 1;
  if (4 * (size_t)m.QW * (K + 1) * sizeof(int) > 64 * 1024) return fail(CL3D_E_UNSUPPORTED, "maxpool_fwd: nsample too large for LDS");
  a.L = m.L; a.QW = m.QW; a.chunks = m.chunks;
  const int tiles_fwd = virtual_tiles(B, ceil_div(M, 4 * m.QW));
  const dim3 grid(tiles_fwd, chunk_grid(tiles_fwd, m.chunks));
  const size_t lds = 4 * (size_t)m.QW * (K + 1) * sizeof(int);
  if (V == 4) hipLaunchKernelGGL((maxpool_fwd_kernel<4>), grid, dim3(256), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((maxpool_fwd_kernel<1>), grid, dim3(256), lds, (hipStream_t)stream, a);
  return check_launch("cl3d_maxpool_fwd");
}

extern "C" int cl3d_maxpool_fwd(const int32_t *idx, const float *ft, int B, int N, int M, int K, int C, float *out,
                                unsigned char *kstar_t, cl3d_stream_t stream) {
  return maxpool_forward(idx, ft, B, N, M, K, C, out, kstar_t, nullptr, stream);
}

extern "C" int cl3d_maxpool_fwd_targets(const int32_t *idx, const float *ft, int B, int N, int M, int K, int C, float *out,
                                        int32_t *target_cm, cl3d_stream_t stream) {
  if (B > 0 && M > 0 && !target_cm) return cl3d::fail(CL3D_E_INVALID, "maxpool_fwd_targets: null pointer");
  return maxpool_forward(idx, ft, B, N, M, K, C, out, nullptr, target_cm, stream);
}

extern "C" int cl3d_maxpool_bwd_targets(const float *gout, const int32_t *target_cm, int B, int N, int M, int C,
                                        float *dfeat, cl3d_stream_t stream) {
  // d features[b, c, i] = sum over the queries j whose arg-max for channel c is support point i of gout[b, c, j]: the
  // PointWiseMLP's arg-max scatter (csrc/fused_pwmlp.hip), every row of d features written (zero where nothing points)
  return cl3d_pwmlp_bwd_hits(gout, target_cm, B, N, M, C, dfeat, stream);
}

extern "C" int cl3d_maxpool_bwd(const float *gout_t, const unsigned char *kstar_t, const int32_t *inv_off,
                                const int32_t *inv_slots, int B, int N, int M, int K, int C, float *dft,
                                int dft_channel_major,
                                cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 0 && N >= 1 && M >= 0 && K >= 1 && K <= 255 && C >= 1, "maxpool_bwd: bad sizes");
  if (B == 0) return CL3D_OK;
  CL3D_REQUIRE(gout_t && kstar_t && inv_off && inv_slots && dft, "maxpool_bwd: null pointer");
  MaxArgs a{};
  a.gout_t = gout_t; a.kstar_t = const_cast<unsigned char *>(kstar_t); a.inv_off = inv_off; a.inv_slots = inv_slots;
  a.dft = dft; a.dft_channel_major = dft_channel_major; a.B = B; a.N = N; a.M = M; a.K = K; a.C = C;
  const int V = (C % 4 == 0) ? 4 : 1;
  const LaneMap m = pick_lane_map(C, V, 64, kMaxpoolWide);
  a.L = m.L; a.QW = m.QW; a.chunks = m.chunks;
  const long long tiles = (long long)B * ceil_div(N, 4 * m.QW);
  const int gx = round_grid(tiles, 8192);
  const dim3 grid(gx, chunk_grid(gx, m.chunks));
  if (V == 4) hipLaunchKernelGGL((maxpool_bwd_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((maxpool_bwd_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("cl3d_maxpool_bwd");
}

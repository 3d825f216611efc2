// fused_pwmlp.hip -- PointWiseMLP ('dp_fi_df', one conv+BN+ReLU layer, max reduction) for gfx950.
//
// Reference (models/local_aggregation_operators.py:288-301): gather [B,C,M,K], build
// x = cat[rel(3), f_centre(C), f_nbr - f_centre(C)] (68.7 MB per cloud at the metric shape), 1x1
// Conv2d (131 -> C_out) over all B*M*K positions (2.2 GFLOP per cloud), BatchNorm2d, ReLU, max over K.
//
// The contraction is linear, so it factors through the points instead of the (point, neighbour) pairs:
//     W x = W_r rel + (W_c - W_d) f_centre + W_d f_nbr
// With G = W_d F and H = (W_c - W_d) F computed ONCE PER POINT (a plain [B*N, C] x [C, 2*C_out] GEMM,
// K times fewer flops than the reference's contraction; done by the caller with a library GEMM so
// autograd also provides dF and dW from dG, dH), every pre-activation is
//     y[b,o,j,k] = W_r[o,:] . rel[b,j,k] + H[b, idx[b,j,0], o] + G[b, idx[b,j,k], o]
// i.e. one point-major row gather + 3 FMAs.  `ght` is [B, N, 2*C_out]: row i = [G_i | H_i].
//
// Nothing of size B*C*M*K is ever stored, and in training the rows are gathered only TWICE (once
// query-major, once support-major), because everything else is algebra on per-(query, channel) values:
//   * BN+ReLU is monotone in y (z = scale*y + shift, sign(scale) = sign(gamma) is known before the batch
//     statistics are), so max_k ReLU(z) = ReLU(z(y*)) with y* = max_k y (min_k y for gamma < 0): the pass
//     that accumulates the batch statistics also finds y* and its slot k*;
//   * the BatchNorm backward is affine in y (dy = A dz [k = k*] + Bc + D y), so sum_k dy and
//     sum_k dy * rel follow from sum_k y, dz and the per-channel sums S_a = sum y*rel_a, R_a = sum rel_a.
// Passes:
//   TRAIN     (gather, query-major) per channel: sum y, sum y^2, S_a, R_a (double partials);
//             per (query, channel): y*, k*, sum_k y;  per slot: slotrec {rel, centre index}
//   APPLY     (element-wise) out = ReLU(scale*y* + shift), transposed to channel-major through LDS
//   FWD       (gather; inference with running statistics) the same output in one pass
//   BWD_ROWS  (element-wise) dz at the arg-max (ReLU gate), d beta = sum dz, d gamma = sum dz*xhat,
//             T_a = sum dz*rel_a(k*) (double partials); then d W_r = A T + Bc R + D S per channel
//   BWD_SUPPORT (gather, support-major through the CSR inverse of idx) dG_i = sum over slots -> i of dy,
//             dH_i = sum over queries centred on i of (D sum_k y + K Bc + A dz).  Ordered, no atomics.
#include "fused_common.h"

namespace cl3d {

enum { PW_TRAIN = 0, PW_FWD = 1 };
constexpr int kSlotBatch = 8;  // row gathers in flight per lane
constexpr int kPartialW = 8;   // doubles per (block, channel) in the partial-sum buffers

struct PwArgs {
  const float *query_xyz, *support_xyz;
  const int *idx;
  const float *ght;  // [B,N,2Co]
  const float *wr;   // [Co,3]
  const float *v0, *v1, *v2;  // per-channel vectors: TRAIN gamma | FWD scale,shift | SUPPORT A,Bc,D
  const float *dzs_in;             // [B,M,Co]
  const unsigned char *kstar_in;   // [B,M,Co]
  const float *sy_in;              // [B,M,Co]
  float *out_t;                    // FWD
  int out_channel_major;
  float *ystar_t, *sy_t;           // TRAIN: [B,M,Co]
  unsigned char *kstar_out;        // TRAIN / FWD
  float4 *slotrec;                 // TRAIN / FWD write {rel, centre index} (may be null); SUPPORT reads
  double *partial;                 // [gridDim.x, Co, kPartialW]
  const int *inv_off, *inv_slots;
  float *dght;                     // SUPPORT: [B,N,2Co]
  int B, N, M, K, Co;
  int L, QW, chunks;
  float inv_radius;
};

__device__ __forceinline__ float pw_preact(const float w[3], float rx, float ry, float rz, float hc, float g) {
  float t = w[0] * rx;
  t = __builtin_fmaf(w[1], ry, t);
  t = __builtin_fmaf(w[2], rz, t);
  return (t + hc) + g;
}

// query-major gather passes.  Persistent blocks: tile = 4*QW queries of one cloud.
// V == 4 is only used when Co % 4 == 0 and V == 1 rows are single elements, so a lane with c0 < Co always
// owns a full vector: every row access is one global_load_dwordx4 / dword.
template <int MODE, int V>
__global__ __launch_bounds__(256) void pwmlp_query_kernel(PwArgs a) {
  extern __shared__ float4 lds4[];
  const int K = a.K, Co = a.Co, M = a.M, N = a.N, L = a.L, QW = a.QW;
  const int TQ = 4 * QW;
  const int row = 2 * Co;
  float4 *slot4 = lds4;  // [TQ*K] {idx, rx, ry, rz}
  double *red = reinterpret_cast<double *>(slot4 + TQ * K);  // [4 waves][L*V*NACC]
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int g = lane / L, cl = lane - g * L;
  const bool lane_on = g < QW;
  const int tiles_per_cloud = (M + TQ - 1) / TQ;
  const int ntiles = a.B * tiles_per_cloud;
  constexpr int NACC = MODE == PW_TRAIN ? 8 : 0;

  for (int ch = 0; ch < a.chunks; ++ch) {
    const int c0 = (ch * L + cl) * V;
    const bool chan_on = lane_on && c0 < Co;
    // sum y, sum y^2, S_0..2 per channel; R_0..2 (sum rel) is the same for every channel of the lane
    constexpr int NCH = NACC > 0 ? 5 : 1;
    double dacc[NCH][V], drel[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < NCH; ++p)
#pragma unroll
      for (int v = 0; v < V; ++v) dacc[p][v] = 0.0;
    float w[V][3], c_v0[V], c_v1[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int c = chan_on ? c0 + v : 0;
      w[v][0] = a.wr[c * 3 + 0];
      w[v][1] = a.wr[c * 3 + 1];
      w[v][2] = a.wr[c * 3 + 2];
      c_v0[v] = a.v0 ? a.v0[c] : 0.f;
      c_v1[v] = a.v1 ? a.v1[c] : 0.f;
    }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      int b, tq;
      decode_tile(tile, a.B, tiles_per_cloud, b, tq);
      const int j0 = tq * TQ;
      const float *q = a.query_xyz + (size_t)b * M * 3;
      const float *s = a.support_xyz + (size_t)b * N * 3;
      __syncthreads();  // previous tile's readers are done with slot4
      for (int t = threadIdx.x; t < TQ * K; t += 256) {
        const int jq = t / K;
        const int j = j0 + jq;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < M) {
          const size_t e = ((size_t)b * M + j) * K + (t - jq * K);
          const int i = a.idx[e];
          r = make_float4(__int_as_float(i), (s[i * 3 + 0] - q[j * 3 + 0]) * a.inv_radius,
                          (s[i * 3 + 1] - q[j * 3 + 1]) * a.inv_radius, (s[i * 3 + 2] - q[j * 3 + 2]) * a.inv_radius);
          // slotrec = {rel, centre index of the query}: one dependent load less per slot in the support-major pass
          if (ch == 0 && a.slotrec != nullptr)
            a.slotrec[e] = make_float4(r.y, r.z, r.w, __int_as_float(a.idx[((size_t)b * M + j) * K]));
        }
        slot4[t] = r;
      }
      __syncthreads();
      const int jq = wave * QW + g;
      const int j = j0 + jq;
      if (!chan_on || j >= M) continue;
      const float4 *myslots = slot4 + jq * K;
      const float *rows = a.ght + (size_t)b * N * row;
      const int ic = __float_as_int(myslots[0].x);  // centre = nearest neighbour (reference :290)
      const Vec<V> hc = load_row<V>(rows + (size_t)ic * row + Co + c0);
      const size_t orow = ((size_t)b * M + j) * Co + c0;

      if constexpr (MODE == PW_TRAIN) {
        float s1[V], s2[V], sr0[V], sr1[V], sr2[V], best[V], sgn[V];
        int kb[V];
        float rs0 = 0.f, rs1 = 0.f, rs2 = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          s1[v] = s2[v] = sr0[v] = sr1[v] = sr2[v] = best[v] = 0.f;
          kb[v] = 0;
          sgn[v] = c_v0[v] < 0.f ? -1.f : 1.f;  // sign(gamma) = sign(scale): which extreme of y wins the max
        }
        for_each_slot<V, kSlotBatch>(myslots, K, rows, row, c0, [&](int k, const float4 &sr, const Vec<V> &gr) {
          rs0 += sr.y;
          rs1 += sr.z;
          rs2 += sr.w;
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const float y = pw_preact(w[v], sr.y, sr.z, sr.w, hc.v[v], gr.v[v]);
            s1[v] += y;
            s2[v] = __builtin_fmaf(y, y, s2[v]);
            sr0[v] = __builtin_fmaf(y, sr.y, sr0[v]);
            sr1[v] = __builtin_fmaf(y, sr.z, sr1[v]);
            sr2[v] = __builtin_fmaf(y, sr.w, sr2[v]);
            const float yy = sgn[v] * y;
            if (k == 0 || yy > best[v]) {  // first extreme
              best[v] = yy;
              kb[v] = k;
            }
          }
        });
        Vec<V> ys, sy;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          ys.v[v] = sgn[v] * best[v];
          sy.v[v] = s1[v];
          dacc[0][v] += (double)s1[v];
          dacc[1][v] += (double)s2[v];
          dacc[2][v] += (double)sr0[v];
          dacc[3][v] += (double)sr1[v];
          dacc[4][v] += (double)sr2[v];
        }
        drel[0] += (double)rs0;
        drel[1] += (double)rs1;
        drel[2] += (double)rs2;
        store_row<V>(a.ystar_t + orow, ys);
        store_row<V>(a.sy_t + orow, sy);
        if constexpr (V == 4) {
          *reinterpret_cast<unsigned *>(a.kstar_out + orow) =
              (unsigned)kb[0] | ((unsigned)kb[1] << 8) | ((unsigned)kb[2] << 16) | ((unsigned)kb[3] << 24);
        } else {
          a.kstar_out[orow] = (unsigned char)kb[0];
        }
      } else {  // PW_FWD
        float best[V];
        int kb[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
          best[v] = 0.f;
          kb[v] = 0;
        }
        for_each_slot<V, kSlotBatch>(myslots, K, rows, row, c0, [&](int k, const float4 &sr, const Vec<V> &gr) {
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const float y = pw_preact(w[v], sr.y, sr.z, sr.w, hc.v[v], gr.v[v]);
            float z = __builtin_fmaf(y, c_v0[v], c_v1[v]);
            z = z > 0.f ? z : 0.f;
            if (k == 0 || z > best[v]) {
              best[v] = z;
              kb[v] = k;
            }
          }
        });
        // the operator's output is channel-major [B,Co,M] at the API boundary: written directly (4-byte
        // stores, merged in L2 across the 16 neighbouring queries of the tile) instead of a transpose pass
        _Pragma("unroll") for (int v = 0; v < V; ++v) {
          if (a.out_channel_major) a.out_t[((size_t)b * Co + c0 + v) * M + j] = best[v];
          else a.out_t[orow + v] = best[v];
          if (a.kstar_out) a.kstar_out[orow + v] = (unsigned char)kb[v];
        }
      }
    }

    if constexpr (NACC > 0) {
      // fixed-order reduction of the double partials of this chunk: lane groups of a wave by shuffles
      // (lanes that own no query hold zeros), the four waves through LDS
      // (lane group gg of the wave = lanes [gg*L, gg*L+L); L need not be a power of two)
      auto fold_groups = [&](double x) {
        double tot = x;
        for (int gg = 1; gg * L + L <= CL3D_WAVE; ++gg) tot += __shfl(x, cl + gg * L, CL3D_WAVE);
        return tot;  // meaningful in group 0
      };
#pragma unroll
      for (int p = 0; p < NCH; ++p)
#pragma unroll
        for (int v = 0; v < V; ++v) dacc[p][v] = fold_groups(dacc[p][v]);
#pragma unroll
      for (int p = 0; p < 3; ++p) drel[p] = fold_groups(drel[p]);
      const int LV = L * V;
      const int slice = LV * NACC;
      __syncthreads();
      if (g == 0) {
        double *mine = red + (size_t)wave * slice + cl * V * NACC;
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
          for (int p = 0; p < NACC; ++p) mine[v * NACC + p] = p < NCH ? dacc[p < NCH ? p : 0][v] : drel[p >= NCH ? p - NCH : 0];
      }
      __syncthreads();
      for (int t = threadIdx.x; t < slice; t += 256) {
        const double sum = ((red[t] + red[slice + t]) + red[2 * slice + t]) + red[3 * slice + t];
        const int c = ch * LV + t / NACC;
        if (c < Co) a.partial[((size_t)blockIdx.x * Co + c) * kPartialW + (t - (t / NACC) * NACC)] = sum;
      }
      __syncthreads();
    }
  }
}

// support-major dense backward pass through the CSR inverse
template <int V>
__global__ __launch_bounds__(256) void pwmlp_support_kernel(PwArgs a) {
  const int K = a.K, Co = a.Co, M = a.M, N = a.N, L = a.L, QW = a.QW;
  const int row = 2 * Co;
  const int MK = M * K;
  const int TR = 4 * QW;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int g = lane / L, cl = lane - g * L;
  if (g >= QW) return;
  const int tiles_per_cloud = (N + TR - 1) / TR;
  const int ntiles = a.B * tiles_per_cloud;
  for (int ch = 0; ch < a.chunks; ++ch) {
    const int c0 = (ch * L + cl) * V;
    if (c0 >= Co) continue;
    float w[V][3], cA[V], cB[V], cD[V], kBc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int c = c0 + v;
      w[v][0] = a.wr[c * 3 + 0];
      w[v][1] = a.wr[c * 3 + 1];
      w[v][2] = a.wr[c * 3 + 2];
      cA[v] = a.v0[c];
      cB[v] = a.v1[c];
      cD[v] = a.v2[c];
      kBc[v] = (float)K * cB[v];
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      int b, tr;
      decode_tile(tile, a.B, tiles_per_cloud, b, tr);
      const int i = tr * TR + wave * QW + g;
      if (i >= N) continue;
      const float *rows = a.ght + (size_t)b * N * row;
      const int *off = a.inv_off + (size_t)b * (N + 1);
      const int *slots = a.inv_slots + (size_t)b * MK;
      const float4 *rec = a.slotrec + (size_t)b * MK;
      const Vec<V> gi = load_row<V>(rows + (size_t)i * row + c0);
      float acc[V], acch[V];
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = acch[v] = 0.f;
      const int s0 = off[i], s1 = off[i + 1];
      const float *dzrow = a.dzs_in + (size_t)b * M * Co + c0;
      const unsigned char *ksrow = a.kstar_in + (size_t)b * M * Co + c0;
      const float *syrow = a.sy_in + (size_t)b * M * Co + c0;
      constexpr int SB = 4;  // slots per batch: 3*SB independent row gathers in flight per lane
      for (int e = s0; e < s1; e += SB) {
        int sl[SB];
        float4 r[SB];
        Vec<V> hc[SB], dz[SB], sqv[SB];
        unsigned ksw[SB];
        bool centre[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) sl[u] = slots[e + u < s1 ? e + u : s1 - 1];
#pragma unroll
        for (int u = 0; u < SB; ++u) r[u] = rec[sl[u]];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int j = sl[u] / K;
          hc[u] = load_row<V>(rows + (size_t)__float_as_int(r[u].w) * row + Co + c0);
          dz[u] = load_row<V>(dzrow + (size_t)j * Co);
          if constexpr (V == 4) ksw[u] = *reinterpret_cast<const unsigned *>(ksrow + (size_t)j * Co);
          else ksw[u] = ksrow[(size_t)j * Co];
          // slot 0 of a query is its centre (reference :290): the rows whose slot list holds (j, 0) are exactly
          // the queries centred on this point, so the centre-feature gradient needs no table of its own.
          // sum_k dy of that query = D sum_k y + K Bc + A dz  (sum_k y was left behind by the forward pass)
#pragma unroll
          for (int v = 0; v < V; ++v) sqv[u].v[v] = 0.f;
          centre[u] = sl[u] - j * K == 0;
          if (centre[u]) sqv[u] = load_row<V>(syrow + (size_t)j * Co);
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          if (e + u >= s1) continue;
          const int k = sl[u] - (sl[u] / K) * K;
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const int ks = (int)((ksw[u] >> (8 * v)) & 0xffu);
            const float y = pw_preact(w[v], r[u].x, r[u].y, r[u].z, hc[u].v[v], gi.v[v]);
            float dy = __builtin_fmaf(cD[v], y, cB[v]);
            dy += (k == ks) ? dz[u].v[v] * cA[v] : 0.f;
            acc[v] += dy;
            if (centre[u]) acch[v] += __builtin_fmaf(cD[v], sqv[u].v[v], __builtin_fmaf(cA[v], dz[u].v[v], kBc[v]));
          }
        }
      }
      float *dst = a.dght + ((size_t)b * N + i) * row + c0;
      _Pragma("unroll") for (int v = 0; v < V; ++v) {
        dst[v] = acc[v];
        dst[Co + v] = acch[v];
      }
    }
  }
}

// ---- element-wise passes over the per-(query, channel) rows ------------------------------------------
// Tile = 64 queries x CW channels (CW a power of two <= 128: Co is walked in its binary decomposition, so a
// thread keeps ONE channel for the whole pass and its partial sums stay in registers).  The channel-major
// side of the transposition ([B,Co,M] output / upstream gradient) goes through an LDS tile, so both sides
// are read and written in full 256-byte rows.
enum { ROWS_APPLY = 0, ROWS_BWD = 1 };
constexpr int kRowsBatch = 4;

struct RowArgs {
  const float *ystar_t;           // [B,M,Co]
  const unsigned char *kstar_t;   // [B,M,Co]
  const float4 *slotrec;          // [B,M,K]
  const float *gout;              // [B,Co,M] (channel-major) or [B,M,Co]
  int gout_channel_major;
  const float *scale, *shift, *mean, *invstd;
  float *out;                     // APPLY: [B,Co,M]
  float *dzs_t;                   // BWD: [B,M,Co]
  double *partial;                // BWD: [gridDim.x, Co, kPartialW]
  int B, M, K, Co;
};

template <int MODE>
__global__ __launch_bounds__(256) void pwmlp_rows_kernel(RowArgs a) {
  __shared__ float tile[128 * 65];
  __shared__ double red[MODE == ROWS_BWD ? 256 * 5 : 1];
  const int M = a.M, Co = a.Co, K = a.K;
  const int tid = threadIdx.x;
  const int tiles_per_cloud = (M + 63) / 64;
  const int ntiles = a.B * tiles_per_cloud;
  for (int cbase = 0; cbase < Co;) {
    int CW = 128;
    while (CW > Co - cbase) CW >>= 1;
    const int cl = tid & (CW - 1), r0 = tid / CW, RS = 256 / CW;
    const int c = cbase + cl;
    const float scale = a.scale[c], shift = a.shift[c];
    float mean = 0.f, invstd = 0.f;
    if constexpr (MODE == ROWS_BWD) {
      mean = a.mean[c];
      invstd = a.invstd[c];
    }
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const int b = t / tiles_per_cloud;
      const int j0 = (t - b * tiles_per_cloud) * 64;
      const int nj = M - j0 < 64 ? M - j0 : 64;
      if constexpr (MODE == ROWS_BWD) {
        if (a.gout_channel_major) {
          for (int e0 = 0; e0 < CW * 64; e0 += 256 * kRowsBatch) {
            float gv[kRowsBatch];
#pragma unroll
            for (int u = 0; u < kRowsBatch; ++u) {
              const int e = e0 + u * 256 + tid;
              const int cc = (e >> 6) < CW ? (e >> 6) : CW - 1, jq = e & 63;
              gv[u] = a.gout[((size_t)b * Co + cbase + cc) * M + j0 + (jq < nj ? jq : nj - 1)];
            }
#pragma unroll
            for (int u = 0; u < kRowsBatch; ++u) {
              const int e = e0 + u * 256 + tid;
              if ((e >> 6) < CW) tile[(e >> 6) * 65 + (e & 63)] = gv[u];
            }
          }
          __syncthreads();
        }
        for (int jb = r0; jb < nj; jb += RS * kRowsBatch) {
          float y[kRowsBatch], gq[kRowsBatch];
          int ks[kRowsBatch];
          float4 rel[kRowsBatch];
#pragma unroll
          for (int u = 0; u < kRowsBatch; ++u) {
            const int jq = jb + u * RS < nj ? jb + u * RS : nj - 1;
            const size_t e = ((size_t)b * M + j0 + jq) * Co + c;
            y[u] = a.ystar_t[e];
            ks[u] = a.kstar_t[e];
            gq[u] = a.gout_channel_major ? tile[cl * 65 + jq] : a.gout[e];
          }
#pragma unroll
          for (int u = 0; u < kRowsBatch; ++u) {
            const int jq = jb + u * RS < nj ? jb + u * RS : nj - 1;
            rel[u] = a.slotrec[((size_t)b * M + j0 + jq) * K + ks[u]];
          }
#pragma unroll
          for (int u = 0; u < kRowsBatch; ++u) {
            const int jq = jb + u * RS;
            if (jq >= nj) continue;
            const float z = __builtin_fmaf(y[u], scale, shift);
            const float dz = z > 0.f ? gq[u] : 0.f;
            a.dzs_t[((size_t)b * M + j0 + jq) * Co + c] = dz;
            acc[0] += (double)dz;
            acc[1] += (double)(dz * ((y[u] - mean) * invstd));
            acc[2] += (double)(dz * rel[u].x);
            acc[3] += (double)(dz * rel[u].y);
            acc[4] += (double)(dz * rel[u].z);
          }
        }
        if (a.gout_channel_major) __syncthreads();  // the tile is overwritten by the next iteration
      } else {
        for (int jb = r0; jb < nj; jb += RS * kRowsBatch) {
          float y[kRowsBatch];
#pragma unroll
          for (int u = 0; u < kRowsBatch; ++u) {
            const int jq = jb + u * RS < nj ? jb + u * RS : nj - 1;
            y[u] = a.ystar_t[((size_t)b * M + j0 + jq) * Co + c];
          }
#pragma unroll
          for (int u = 0; u < kRowsBatch; ++u) {
            const int jq = jb + u * RS;
            if (jq >= nj) continue;
            const float z = __builtin_fmaf(y[u], scale, shift);
            tile[cl * 65 + jq] = z > 0.f ? z : 0.f;
          }
        }
        __syncthreads();
        for (int e = tid; e < CW * 64; e += 256) {
          const int cc = e >> 6, jq = e & 63;
          if (jq < nj) a.out[((size_t)b * Co + cbase + cc) * M + j0 + jq] = tile[cc * 65 + jq];
        }
        __syncthreads();
      }
    }
    if constexpr (MODE == ROWS_BWD) {  // fixed-order reduction over the RS threads that share a channel
#pragma unroll
      for (int p = 0; p < 5; ++p) red[tid * 5 + p] = acc[p];
      __syncthreads();
      for (int e = tid; e < CW * 5; e += 256) {
        const int cc = e / 5, p = e - cc * 5;
        double sum = 0.0;
        for (int r = 0; r < RS; ++r) sum += red[(r * CW + cc) * 5 + p];
        a.partial[((size_t)blockIdx.x * Co + cbase + cc) * kPartialW + p] = sum;
      }
      __syncthreads();
    }
    cbase += CW;
  }
}

// ---- fixed-order reduction of the per-block double partials + the per-channel BatchNorm algebra.
// One block per channel; replaces ~30 tiny element-wise launches the same math costs in PyTorch.
enum { FIN_STATS = 0, FIN_COEFFS = 1 };

struct FinArgs {
  const double *partial;  // [G, Co, kPartialW]
  int G, Co;
  double count;
  float eps, momentum;
  const float *gamma, *beta, *mean_in, *invstd_in;
  float *running_mean, *running_var;
  double *sums;           // [Co,6]: S_a = sum y*rel_a, R_a = sum rel_a  (STATS writes, COEFFS reads)
  float *o0, *o1, *o2, *o3, *o4, *o5;
};

template <int MODE>
__global__ __launch_bounds__(256) void pwmlp_finalize_kernel(FinArgs a) {
  constexpr int NS = MODE == FIN_STATS ? 8 : 5;
  __shared__ double s_red[NS][256];
  const int c = blockIdx.x;
  double acc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) acc[k] = 0.0;
  for (int g = threadIdx.x; g < a.G; g += 256) {
    const double *p = a.partial + ((size_t)g * a.Co + c) * kPartialW;
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] += p[k];
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) s_red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) {
#pragma unroll
      for (int k = 0; k < NS; ++k) s_red[k][threadIdx.x] += s_red[k][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  if constexpr (MODE == FIN_STATS) {
    const double s0 = s_red[0][0], s1 = s_red[1][0];
    const double mean = s0 / a.count;
    double var = s1 / a.count - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double invstd = 1.0 / sqrt(var + (double)a.eps);
    const double scale = (double)a.gamma[c] * invstd;
    a.o0[c] = (float)scale;
    a.o1[c] = (float)((double)a.beta[c] - mean * scale);
    a.o2[c] = (float)mean;
    a.o3[c] = (float)invstd;
#pragma unroll
    for (int k = 0; k < 6; ++k) a.sums[c * 6 + k] = s_red[2 + k][0];
    if (a.running_mean != nullptr) {  // nn.BatchNorm2d: running = (1-m) running + m batch, unbiased variance
      const double unbiased = var * (a.count / (a.count > 1.0 ? a.count - 1.0 : 1.0));
      a.running_mean[c] = a.running_mean[c] * (1.0f - a.momentum) + a.momentum * (float)mean;
      a.running_var[c] = a.running_var[c] * (1.0f - a.momentum) + a.momentum * (float)unbiased;
    }
  } else {
    // BatchNorm backward, affine in y:  dy = A dz [k = k*] + Bc + D y   (s0 = sum dz = d beta, s1 = sum dz*xhat = d gamma)
    const double s0 = s_red[0][0], s1 = s_red[1][0];
    const double invstd = (double)a.invstd_in[c], mean = (double)a.mean_in[c];
    const double A = (double)a.gamma[c] * invstd;
    const double D = -A * invstd * s1 / a.count;
    const double Bc = -A * s0 / a.count - D * mean;
    a.o0[c] = (float)A;
    a.o1[c] = (float)Bc;
    a.o2[c] = (float)D;
    a.o3[c] = (float)s1;  // d gamma
    a.o4[c] = (float)s0;  // d beta
    // d W_r[c][a] = sum_{slots} dy * rel_a = A T_a + Bc R_a + D S_a
#pragma unroll
    for (int k = 0; k < 3; ++k)
      a.o5[c * 3 + k] = (float)(A * s_red[2 + k][0] + Bc * a.sums[c * 6 + 3 + k] + D * a.sums[c * 6 + k]);
  }
}

static int pw_check(const PwArgs &a, const char *who) {
  if (a.B < 0 || a.N < 1 || a.M < 1 || a.K < 1 || a.Co < 1) return fail(CL3D_E_INVALID, "%s: bad sizes", who);
  if (a.K > 255) return fail(CL3D_E_UNSUPPORTED, "%s: nsample=%d > 255 (arg-max is stored in a byte)", who, a.K);
  if ((long long)a.M * a.K > 0x7fffffffLL) return fail(CL3D_E_UNSUPPORTED, "%s: M*K too large", who);
  return CL3D_OK;
}

static LaneMap pw_lane_map(int Co, int K, int V, int nacc, size_t *lds_out) {
  LaneMap m = pick_lane_map(Co, V);
  if (m.QW > 16) {
    m.QW = 16;
    m.L = 4;
    m.chunks = ((Co + V - 1) / V + m.L - 1) / m.L;
  }
  for (;;) {
    const size_t tq = 4 * (size_t)m.QW;
    const size_t lds = tq * K * sizeof(float4) + (size_t)4 * m.L * V * nacc * sizeof(double);
    if (lds <= 60 * 1024 || m.QW == 1) {
      *lds_out = lds;
      return m;
    }
    m.QW -= 1;
  }
}

template <int MODE>
static int launch_query(PwArgs &a, int nacc, int n_partials, hipStream_t st, const char *who) {
  const int V = (a.Co % 4 == 0) ? 4 : 1;
  size_t lds = 0;
  const LaneMap m = pw_lane_map(a.Co, a.K, V, nacc, &lds);
  if (lds > 64 * 1024) return fail(CL3D_E_UNSUPPORTED, "%s: nsample=%d needs %zu B of LDS", who, a.K, lds);
  a.L = m.L; a.QW = m.QW; a.chunks = m.chunks;
  const long long tiles = (long long)a.B * ceil_div(a.M, 4 * m.QW);
  const int gx = nacc > 0 ? n_partials : round_grid(tiles, 8192);
  if (V == 4) hipLaunchKernelGGL((pwmlp_query_kernel<MODE, 4>), dim3(gx), dim3(256), lds, st, a);
  else hipLaunchKernelGGL((pwmlp_query_kernel<MODE, 1>), dim3(gx), dim3(256), lds, st, a);
  return check_launch(who);
}

}  // namespace cl3d

extern "C" int cl3d_pwmlp_partials(int B, int M, int Co) {
  (void)Co;
  return cl3d::round_grid(((long long)B * M + 15) / 16, 1024);
}

extern "C" int cl3d_pwmlp_stats(const float *query_xyz, const float *support_xyz, const int32_t *idx,
                                const float *ght, const float *wr, const float *gamma, int B, int N, int M,
                                int K, int Co, float radius, float *ystar_t, unsigned char *kstar_t,
                                float *sy_t, float *slotrec, double *partial, int n_partials,
                                cl3d_stream_t stream) {
  using namespace cl3d;
  PwArgs a{};
  a.query_xyz = query_xyz; a.support_xyz = support_xyz; a.idx = idx; a.ght = ght; a.wr = wr; a.v0 = gamma;
  a.ystar_t = ystar_t; a.kstar_out = kstar_t; a.sy_t = sy_t; a.slotrec = reinterpret_cast<float4 *>(slotrec);
  a.partial = partial;
  a.B = B; a.N = N; a.M = M; a.K = K; a.Co = Co; a.inv_radius = 1.0f / radius;
  int rc = pw_check(a, "pwmlp_stats");
  if (rc != CL3D_OK) return rc;
  CL3D_REQUIRE(query_xyz && support_xyz && idx && ght && wr && gamma && ystar_t && kstar_t && sy_t && partial,
               "pwmlp_stats: null pointer");
  CL3D_REQUIRE(n_partials == cl3d_pwmlp_partials(B, M, Co), "pwmlp_stats: partial buffer must have cl3d_pwmlp_partials() blocks");
  if (B == 0) return CL3D_OK;
  return launch_query<PW_TRAIN>(a, 8, n_partials, (hipStream_t)stream, "cl3d_pwmlp_stats");
}

extern "C" int cl3d_pwmlp_finalize_stats(const double *partial, int n_partials, int Co, double count, float eps,
                                         float momentum, const float *gamma, const float *beta,
                                         float *running_mean, float *running_var, float *scale, float *shift,
                                         float *mean, float *invstd, double *sums, cl3d_stream_t stream) {
  CL3D_REQUIRE(partial && gamma && beta && scale && shift && mean && invstd && sums && n_partials > 0 && Co > 0 && count > 0,
               "pwmlp_finalize_stats: bad arguments");
  cl3d::FinArgs a{};
  a.partial = partial; a.G = n_partials; a.Co = Co; a.count = count; a.eps = eps; a.momentum = momentum;
  a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var;
  a.o0 = scale; a.o1 = shift; a.o2 = mean; a.o3 = invstd; a.sums = sums;
  hipLaunchKernelGGL((cl3d::pwmlp_finalize_kernel<cl3d::FIN_STATS>), dim3(Co), dim3(256), 0, (hipStream_t)stream, a);
  return cl3d::check_launch("cl3d_pwmlp_finalize_stats");
}

extern "C" int cl3d_pwmlp_apply(const float *ystar_t, const float *scale, const float *shift, int B, int M,
                                int Co, float *out, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 0 && M >= 1 && Co >= 1, "pwmlp_apply: bad sizes");
  CL3D_REQUIRE(ystar_t && scale && shift && out, "pwmlp_apply: null pointer");
  if (B == 0) return CL3D_OK;
  RowArgs a{};
  a.ystar_t = ystar_t; a.scale = scale; a.shift = shift; a.out = out; a.B = B; a.M = M; a.Co = Co; a.K = 1;
  const int gx = round_grid((long long)B * ceil_div(M, 64), 8192);
  hipLaunchKernelGGL((pwmlp_rows_kernel<ROWS_APPLY>), dim3(gx), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("cl3d_pwmlp_apply");
}

extern "C" int cl3d_pwmlp_bn_backward_coeffs(const double *partial, int n_partials, int Co, double count,
                                             const float *gamma, const float *mean, const float *invstd,
                                             const double *sums, float *cA, float *cB, float *cD, float *dgamma,
                                             float *dbeta, float *dwr, cl3d_stream_t stream) {
  CL3D_REQUIRE(partial && gamma && mean && invstd && sums && cA && cB && cD && dgamma && dbeta && dwr && n_partials > 0 &&
                   Co > 0 && count > 0,
               "pwmlp_bn_backward_coeffs: bad arguments");
  cl3d::FinArgs a{};
  a.partial = partial; a.G = n_partials; a.Co = Co; a.count = count; a.gamma = gamma; a.mean_in = mean;
  a.invstd_in = invstd; a.sums = const_cast<double *>(sums);
  a.o0 = cA; a.o1 = cB; a.o2 = cD; a.o3 = dgamma; a.o4 = dbeta; a.o5 = dwr;
  hipLaunchKernelGGL((cl3d::pwmlp_finalize_kernel<cl3d::FIN_COEFFS>), dim3(Co), dim3(256), 0, (hipStream_t)stream, a);
  return cl3d::check_launch("cl3d_pwmlp_bn_backward_coeffs");
}

extern "C" int cl3d_pwmlp_fwd(const float *query_xyz, const float *support_xyz, const int32_t *idx,
                              const float *ght, const float *wr, const float *scale, const float *shift,
                              int B, int N, int M, int K, int Co, float radius, float *out,
                              int out_channel_major, unsigned char *kstar_t, float *slotrec,
                              cl3d_stream_t stream) {
  using namespace cl3d;
  PwArgs a{};
  a.query_xyz = query_xyz; a.support_xyz = support_xyz; a.idx = idx; a.ght = ght; a.wr = wr;
  a.v0 = scale; a.v1 = shift; a.out_t = out; a.out_channel_major = out_channel_major; a.kstar_out = kstar_t;
  a.slotrec = reinterpret_cast<float4 *>(slotrec);
  a.B = B; a.N = N; a.M = M; a.K = K; a.Co = Co; a.inv_radius = 1.0f / radius;
  int rc = pw_check(a, "pwmlp_fwd");
  if (rc != CL3D_OK) return rc;
  CL3D_REQUIRE(query_xyz && support_xyz && idx && ght && wr && scale && shift && out, "pwmlp_fwd: null pointer");
  if (B == 0) return CL3D_OK;
  return launch_query<PW_FWD>(a, 0, 0, (hipStream_t)stream, "cl3d_pwmlp_fwd");
}

extern "C" int cl3d_pwmlp_bwd_rows(const float *gout, int gout_channel_major, const float *ystar_t,
                                   const unsigned char *kstar_t, const float *slotrec, const float *scale,
                                   const float *shift, const float *mean, const float *invstd, int B, int M,
                                   int K, int Co, float *dzs_t, double *partial, int n_partials,
                                   cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 0 && M >= 1 && K >= 1 && K <= 255 && Co >= 1, "pwmlp_bwd_rows: bad sizes");
  CL3D_REQUIRE(gout && ystar_t && kstar_t && slotrec && scale && shift && mean && invstd && dzs_t && partial,
               "pwmlp_bwd_rows: null pointer");
  CL3D_REQUIRE(n_partials == cl3d_pwmlp_partials(B, M, Co), "pwmlp_bwd_rows: wrong partial block count");
  if (B == 0) return CL3D_OK;
  RowArgs a{};
  a.gout = gout; a.gout_channel_major = gout_channel_major; a.ystar_t = ystar_t; a.kstar_t = kstar_t;
  a.slotrec = reinterpret_cast<const float4 *>(slotrec); a.scale = scale; a.shift = shift; a.mean = mean;
  a.invstd = invstd; a.dzs_t = dzs_t; a.partial = partial; a.B = B; a.M = M; a.K = K; a.Co = Co;
  hipLaunchKernelGGL((pwmlp_rows_kernel<ROWS_BWD>), dim3(n_partials), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("cl3d_pwmlp_bwd_rows");
}

extern "C" int cl3d_pwmlp_bwd_support(const int32_t *idx, const float *ght, const float *wr, const float *cA,
                                      const float *cB, const float *cD, const float *dzs_t,
                                      const unsigned char *kstar_t, const float *slotrec, const float *sy_t,
                                      const int32_t *inv_off, const int32_t *inv_slots, int B, int N, int M,
                                      int K, int Co, float *dght, cl3d_stream_t stream) {
  using namespace cl3d;
  PwArgs a{};
  a.idx = idx; a.ght = ght; a.wr = wr; a.v0 = cA; a.v1 = cB; a.v2 = cD; a.dzs_in = dzs_t; a.kstar_in = kstar_t;
  a.slotrec = reinterpret_cast<float4 *>(const_cast<float *>(slotrec)); a.sy_in = sy_t;
  a.inv_off = inv_off; a.inv_slots = inv_slots; a.dght = dght;
  a.B = B; a.N = N; a.M = M; a.K = K; a.Co = Co;
  int rc = pw_check(a, "pwmlp_bwd_support");
  if (rc != CL3D_OK) return rc;
  CL3D_REQUIRE(idx && ght && wr && cA && cB && cD && dzs_t && kstar_t && slotrec && sy_t && inv_off && inv_slots && dght,
               "pwmlp_bwd_support: null pointer");
  if (B == 0) return CL3D_OK;
  const int V = (Co % 4 == 0) ? 4 : 1;
  const LaneMap m = pick_lane_map(Co, V);
  a.L = m.L; a.QW = m.QW; a.chunks = m.chunks;
  const long long tiles = (long long)B * ceil_div(N, 4 * m.QW);
  const int gx = round_grid(tiles, 8192);
  if (V == 4) hipLaunchKernelGGL((pwmlp_support_kernel<4>), dim3(gx), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((pwmlp_support_kernel<1>), dim3(gx), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("cl3d_pwmlp_bwd_support");
}

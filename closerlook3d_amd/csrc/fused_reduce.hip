// fused_reduce.hip -- PosPool / AdaptiveWeight / PseudoGrid without the [B,C,M,K] tensor (gfx950).
//
// All three operators of the reference have the shape
//     out[b,c,j] = reduce_k  w_c(rel[b,j,k]) * mask[b,j,k] * f[b,c,idx[b,j,k]]
// (models/local_aggregation_operators.py: PosPool :65-103, AdaptiveWeight :188-214, PseudoGrid
// :383-419) and differ only in the positional weight w_c.  The reference materialises the grouped
// features (33.5 MB per cloud at the metric shape) and runs 6-10 element-wise passes over them; here
// one kernel reads each neighbour row once (point-major rows, see fused_common.h) and keeps the
// K-reduction in registers.  The backward kernel is a gather as well, through the CSR inverse of idx
// (csr.hip): d f[b,i,:] = sum over the slots referencing i, in ascending slot order -- no atomics.
//
//   OP_POSPOOL_XYZ     w_c = rel[c % 3]
//   OP_POSPOOL_SINCOS  c = a*2fd + s*fd + f:  w_c = s ? cos : sin ((100*rel_a) / dim[f])
//   OP_ADAPTIVE        w_c = bias[c/S] + W[c/S,:] . rel          (weight_type 'dp', one conv layer)
//   OP_PSEUDOGRID      out_c = sum_p kw[p,c] * sum_k h_p(rel_k) mask_k f_c,k,  h_p = max(1 - |rel-KP_p| / extent, 0)
#include "fused_common.h"

namespace cl3d {

enum { OP_POSPOOL_XYZ = 0, OP_POSPOOL_SINCOS = 1, OP_ADAPTIVE = 2, OP_PSEUDOGRID = 3 };
enum { RED_SUM = 0, RED_AVG = 1 };
constexpr int kMaxKP = 16;  // kernel points per PseudoGrid operator (reference default 15)

struct ReduceArgs {
  const float *query_xyz, *support_xyz;
  const int *query_mask, *idx, *idx_mask;
  const float *ft;       // [B,N,C]
  const float *gout_t;   // bwd: [B,M,C]
  const float *p0, *p1;  // operator parameters (see table above)
  float *out_t;          // fwd: [B,M,C]
  float4 *slotrec;       // [B,M,K]  {rel.x, rel.y, rel.z, coef}; fwd writes (may be null), bwd reads
  float4 *pairs;         // [B,M,K,2] PseudoGrid (C % 4 == 0): the slot's non-zero influences, see PgPairs; fwd writes, bwd reads
  const int *inv_off, *inv_slots;
  float *dft;            // bwd: [B,N,C], or [B,C,N] when dft_channel_major
  int dft_channel_major;
  float *dparam;         // bwd: [gridDim.x, C, NP] partial parameter gradients (may be null)
  int B, N, M, K, C;
  int L, QW, chunks;
  int reduction, normalize, pint;  // pint: S (adaptive) / P (pseudo grid)
  int constant_influence, out_channel_major;
  float inv_radius, pfloat;        // pfloat: 1/extent (pseudo grid)
};

// Packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of fp32 per VALU slot).  PseudoGrid spends
// P x V = 15 x 4 FMAs per (slot, lane); with the V = 4 channels of a lane held as two float pairs the same
// arithmetic (each product and sum rounded exactly as before) issues half as many instructions.
// CL3D_PG_PK = 0: the same pairs as two scalar FMAs each (scripts/micro/kernel_variants.py "pg_scalar").  Measured in round 6,
// session 67, because the TRAIN walk and the ball query got FASTER as scalar code: here the packed form stays -- the operator
// step 0.3896-0.3904 against 0.3927-0.3935 ms, config 3 4.63-4.66 against 4.69-4.70, every pair (15 influences x 4 channels per
// slot: the instruction count matters).  Its packed operands carry `op_sel` but come out of VALU arithmetic, not
// out of ds_read_b128, and the kernels stayed bit-exact beside bf16 contractions in every survey (up to 400 launches).
#ifndef CL3D_PG_PK
#define CL3D_PG_PK 1
#endif
#if CL3D_PG_PK
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 pk_splat(float x) { return (f2)(x); }
#else
struct f2 {
  float e[2];
  __device__ __forceinline__ float &operator[](int i) { return e[i]; }
  __device__ __forceinline__ const float &operator[](int i) const { return e[i]; }
};
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) {
  f2 r;
  r.e[0] = __builtin_fmaf(a.e[0], b.e[0], c.e[0]);
  r.e[1] = __builtin_fmaf(a.e[1], b.e[1], c.e[1]);
  return r;
}
__device__ __forceinline__ f2 pk_splat(float x) {
  f2 r;
  r.e[0] = x;
  r.e[1] = x;
  return r;
}
#endif

__device__ __forceinline__ float kp_influence(float rx, float ry, float rz, const float *kp, float inv_extent,
                                              int constant) {
  if (constant) return 1.0f;
  const float dx = rx - kp[0], dy = ry - kp[1], dz = rz - kp[2];
  const float sq = dx * dx + dy * dy + dz * dz;
  // v_sqrt_f32 (1 ulp) instead of the IEEE-exact sequence: 15 of these per slot, three passes per step, and the
  // influence feeds sums compared at 1e-5
  const float h = 1.0f - __builtin_amdgcn_sqrtf(sq) * inv_extent;
  return h > 0.0f ? h : 0.0f;
}

// PseudoGrid, sparse form.  A kernel point only influences neighbours closer than `extent` to it, and the kernel points
// sit ~1.5 extents apart: at the reference's settings a slot has 1.2 non-zero influences on average out of 15 (19 % of
// the slots have none, 0.5 % have four).  So a slot is reduced to at most kPgPairs (kernel point, influence * mask)
// pairs, evaluated ONCE (by the forward pass's staging threads) and kept next to slotrec for the backward passes, and
// the slot's per-channel weight is formed from them,
//     out_c = sum_k (sum_{(p,h) in pairs(k)} h * kw[p,c]) * f_c[idx_k],
// with kw rows read from an LDS table: ~8 packed FMAs per (slot, lane) instead of 32.  Unused pairs hold h = 0, p = 0
// (they add 0 * kw[0]); a slot with more than kPgPairs influences (constant influence: all of them) is flagged
// p[3] = -1 and takes the dense sum.  Same terms as the reference's sum over kernel points, added per slot instead of
// per kernel point.
constexpr int kPgPairs = 4;

// the staging side: influences of one slot -> its pair record in LDS (hq, pq zero-initialised by the caller)
__device__ __forceinline__ void pg_stage_pairs(const ReduceArgs &a, float rx, float ry, float rz, float m, float4 *hq,
                                               int4 *pq) {
  float *hl = reinterpret_cast<float *>(hq);
  int *pl = reinterpret_cast<int *>(pq);
  int n = 0;
#pragma unroll
  for (int p = 0; p < kMaxKP; ++p) {
    if (p >= a.pint) break;
    const float h = kp_influence(rx, ry, rz, a.p0 + p * 3, a.pfloat, a.constant_influence) * m;
    if (h != 0.f) {
      if (n < kPgPairs) {
        hl[n] = h;
        pl[n] = p;
      }
      ++n;
    }
  }
  if (n > kPgPairs) pl[kPgPairs - 1] = -1;
}

// the consuming side: weight of the lane's four channels for one slot.  kwrow = the lane's column of the LDS table
// kwl [kMaxKP][LV] (row stride LV floats).  `rel` / `m` are only read on the dense path.
__device__ __forceinline__ void pg_slot_weight(const ReduceArgs &a, const float4 hv, const int4 pv, const float *kwrow,
                                               int LV, float rx, float ry, float rz, float m, f2 &wlo, f2 &whi) {
  wlo = pk_splat(0.f);
  whi = pk_splat(0.f);
  if (__builtin_expect(__ballot(pv.w < 0) != 0ull, 0)) {  // some lane group's slot has more than kPgPairs influences
    if (pv.w < 0) {
      for (int p = 0; p < a.pint; ++p) {
        const float h = kp_influence(rx, ry, rz, a.p0 + p * 3, a.pfloat, a.constant_influence) * m;
        const float4 kw = *reinterpret_cast<const float4 *>(kwrow + p * LV);
        wlo = pk_fma(pk_splat(h), (f2){kw.x, kw.y}, wlo);
        whi = pk_fma(pk_splat(h), (f2){kw.z, kw.w}, whi);
      }
      return;
    }
  }
  const float4 k0 = *reinterpret_cast<const float4 *>(kwrow + pv.x * LV);
  const float4 k1 = *reinterpret_cast<const float4 *>(kwrow + pv.y * LV);
  wlo = pk_fma(pk_splat(hv.x), (f2){k0.x, k0.y}, wlo);
  whi = pk_fma(pk_splat(hv.x), (f2){k0.z, k0.w}, whi);
  wlo = pk_fma(pk_splat(hv.y), (f2){k1.x, k1.y}, wlo);
  whi = pk_fma(pk_splat(hv.y), (f2){k1.z, k1.w}, whi);
  if (__ballot(hv.z != 0.f) != 0ull) {  // a third / fourth pair somewhere in the wave (one slot in twenty has one)
    const int p2 = pv.w < 0 ? 0 : pv.z, p3 = pv.w < 0 ? 0 : pv.w;
    const float4 k2 = *reinterpret_cast<const float4 *>(kwrow + p2 * LV);
    const float4 k3 = *reinterpret_cast<const float4 *>(kwrow + p3 * LV);
    wlo = pk_fma(pk_splat(hv.z), (f2){k2.x, k2.y}, wlo);
    whi = pk_fma(pk_splat(hv.z), (f2){k2.z, k2.w}, whi);
    wlo = pk_fma(pk_splat(hv.w), (f2){k3.x, k3.y}, wlo);
    whi = pk_fma(pk_splat(hv.w), (f2){k3.z, k3.w}, whi);
  }
}

// per-lane description of how its V channels turn a relative position into a weight
template <int OP, int V>
struct ChannelWeights {
  int axis[V];     // xyz / sincos: which coordinate
  int is_cos[V];   // sincos
  float dim[V];    // sincos divisor
  float w[V][3];   // adaptive: conv weight row
  float bias[V];   // adaptive
  __device__ __forceinline__ void init(const ReduceArgs &a, int c0) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int c = c0 + v;
      if constexpr (OP == OP_POSPOOL_XYZ) {
        axis[v] = c % 3;
      } else if constexpr (OP == OP_POSPOOL_SINCOS) {
        const int fd = a.C / 6;
        axis[v] = c / (2 * fd);
        const int rem = c - axis[v] * 2 * fd;
        is_cos[v] = rem / fd;
        dim[v] = a.p0[rem - is_cos[v] * fd];
      } else if constexpr (OP == OP_ADAPTIVE) {
        const int cw = c / a.pint;
        w[v][0] = a.p0[cw * 3 + 0];
        w[v][1] = a.p0[cw * 3 + 1];
        w[v][2] = a.p0[cw * 3 + 2];
        bias[v] = a.p1[cw];
      }
    }
  }
  __device__ __forceinline__ float weight(int v, float rx, float ry, float rz) const {
    if constexpr (OP == OP_POSPOOL_XYZ) {
      return axis[v] == 0 ? rx : (axis[v] == 1 ? ry : rz);
    } else if constexpr (OP == OP_POSPOOL_SINCOS) {
      const float r = axis[v] == 0 ? rx : (axis[v] == 1 ? ry : rz);
      const float arg = (100.0f * r) / dim[v];
      return is_cos[v] ? cosf(arg) : sinf(arg);
    } else if constexpr (OP == OP_ADAPTIVE) {
      return bias[v] + w[v][0] * rx + w[v][1] * ry + w[v][2] * rz;
    } else {
      return 0.0f;
    }
  }
};

// -------------------------------------------------------------------------------- forward
template <int OP, int V, bool SPARSE = false>  // SPARSE: PseudoGrid with linear influence, C % 4 == 0 (pair records)
__global__ __launch_bounds__(256) void fused_reduce_fwd_kernel(ReduceArgs a) {
  extern __shared__ float4 lds4[];
  const int K = a.K, C = a.C, M = a.M, N = a.N, L = a.L, QW = a.QW;
  const int TQ = 4 * QW;
  // per-query rows are K+1 long: the lane groups of a wave read the same slot of different queries at once, and
  // rows of K float4 (512 B at K = 32) would put them all in the same LDS banks
  const int KS = K + 1;
  float4 *slot4 = lds4;                                    // [TQ][KS] {idx, rx, ry, rz}
  // PseudoGrid: sparse form: the slots' pair records, plane 0 = influences, plane 1 = kernel points (pg_stage_pairs);
  //             dense form (C % 4 != 0, or constant influence where every slot sees every kernel point):
  //             [kMaxKP/4][TQ][KS] influences, 4 kernel points each
  constexpr bool PG_SPARSE = OP == OP_PSEUDOGRID && V == 4 && SPARSE;
  constexpr int kPgPlanes = OP == OP_PSEUDOGRID ? (PG_SPARSE ? 2 : kMaxKP / 4) : 0;
  float4 *hbuf4 = slot4 + TQ * KS;
  float *coef = reinterpret_cast<float *>(hbuf4 + kPgPlanes * TQ * KS);  // [TQ][KS] mask weight
  float *cntq = coef + TQ * KS;                            // [TQ]
  float *kwl = cntq + TQ;                                  // PG_SPARSE: [kMaxKP][L*V] kernel weights of the channel chunk
  int b, tq;
  decode_tile(blockIdx.x, a.B, (M + TQ - 1) / TQ, b, tq);
  const int j0 = tq * TQ;
  const float *q = a.query_xyz + (size_t)b * M * 3;
  const float *s = a.support_xyz + (size_t)b * N * 3;

  // ---- phase A: per-slot scalars, once per block
  for (int t = threadIdx.x; t < TQ * K; t += 256) {
    const int jq = t / K;
    const int j = j0 + jq;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    float m = 0.f;
    if (j < M) {
      const size_t e = ((size_t)b * M + j) * K + (t - jq * K);
      const int i = a.idx[e];
      m = (float)(a.idx_mask[e] + (1 - a.query_mask[(size_t)b * M + j]));
      float dx = s[i * 3 + 0] - q[j * 3 + 0];
      float dy = s[i * 3 + 1] - q[j * 3 + 1];
      float dz = s[i * 3 + 2] - q[j * 3 + 2];
      if (a.normalize) {
        dx *= a.inv_radius;
        dy *= a.inv_radius;
        dz *= a.inv_radius;
      }
      r = make_float4(__int_as_float(i), dx, dy, dz);
    }
    const int ts = jq * KS + (t - jq * K);
    if constexpr (PG_SPARSE) {
      hbuf4[ts] = make_float4(0.f, 0.f, 0.f, 0.f);
      hbuf4[TQ * KS + ts] = make_float4(0.f, 0.f, 0.f, 0.f);  // kernel point 0 four times
      if (j < M) pg_stage_pairs(a, r.y, r.z, r.w, m, hbuf4 + ts, reinterpret_cast<int4 *>(hbuf4 + TQ * KS + ts));
    } else if constexpr (OP == OP_PSEUDOGRID) {
      float h[kMaxKP];
#pragma unroll
      for (int p = 0; p < kMaxKP; ++p)
        h[p] = (j < M && p < a.pint) ? kp_influence(r.y, r.z, r.w, a.p0 + p * 3, a.pfloat, a.constant_influence) * m : 0.f;
#pragma unroll
      for (int p4 = 0; p4 < kMaxKP / 4; ++p4)
        hbuf4[p4 * TQ * KS + ts] = make_float4(h[4 * p4], h[4 * p4 + 1], h[4 * p4 + 2], h[4 * p4 + 3]);
    }
    slot4[ts] = r;
    coef[ts] = m;
  }
  __syncthreads();
  if ((int)threadIdx.x < TQ) {
    float n = 0.f;
    for (int k = 0; k < K; ++k) n += coef[threadIdx.x * KS + k];
    cntq[threadIdx.x] = n;
  }
  __syncthreads();
  if (a.slotrec != nullptr && blockIdx.y == 0) {
    for (int t = threadIdx.x; t < TQ * K; t += 256) {
      const int jq = t / K;
      const int j = j0 + jq;
      if (j < M) {
        const int ts = jq * KS + (t - jq * K);
        const float4 r = slot4[ts];
        const float cf = a.reduction == RED_AVG ? coef[ts] / cntq[jq] : coef[ts];
        a.slotrec[((size_t)b * M + j) * K + (t - jq * K)] = make_float4(r.y, r.z, r.w, cf);
        if constexpr (PG_SPARSE) {
          if (a.pairs != nullptr) {
            float4 *dst = a.pairs + (((size_t)b * M + j) * K + (t - jq * K)) * 2;
            dst[0] = hbuf4[ts];
            dst[1] = hbuf4[TQ * KS + ts];
          }
        }
      }
    }
  }

  // ---- phase B: one lane group per query, K-reduction in registers
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int g = lane / L, cl = lane - g * L;
  const int jq = g < QW ? wave * QW + g : 0;
  const int j = j0 + jq;
  const bool active = g < QW && j < M;
  if (!PG_SPARSE && !active) return;  // (the sparse PseudoGrid path has block barriers in the chunk loop)
  const float n = cntq[jq];
  const float4 *myslots = slot4 + jq * KS;
  const float *mycoef = coef + jq * KS;
  const float *frow = a.ft + (size_t)b * N * C;
  for (int ch = blockIdx.y; ch < a.chunks; ch += gridDim.y) {
    const int c0 = (ch * L + cl) * V;
    if constexpr (PG_SPARSE) {
      const int LV = L * V;
      __syncthreads();  // the previous chunk's readers are done with the table
      for (int t = threadIdx.x; t < kMaxKP * LV; t += 256) {
        const int p = t / LV, c = ch * LV + (t - p * LV);
        kwl[t] = (p < a.pint && c < C) ? a.p1[(size_t)p * C + c] : 0.f;
      }
      __syncthreads();
      if (!active || c0 >= C) continue;
      const float *kwrow = kwl + cl * V;
      const float4 *hq = hbuf4 + jq * KS;
      const int4 *pq = reinterpret_cast<const int4 *>(hbuf4 + TQ * KS + jq * KS);
      f2 alo = pk_splat(0.f), ahi = pk_splat(0.f);
      constexpr int KB = 4;  // row gathers in flight per lane
      for (int k0 = 0; k0 < K; k0 += KB) {
        Vec<V> f[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u)
          f[u] = load_row<V>(frow + (size_t)__float_as_int(myslots[k0 + u < K ? k0 + u : K - 1].x) * C + c0);
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          if (k0 + u >= K) continue;
          const float4 sr = myslots[k0 + u];
          f2 wlo, whi;
          pg_slot_weight(a, hq[k0 + u], pq[k0 + u], kwrow, LV, sr.y, sr.z, sr.w, mycoef[k0 + u], wlo, whi);
          alo = pk_fma(wlo, (f2){f[u].v[0], f[u].v[1]}, alo);
          ahi = pk_fma(whi, (f2){f[u].v[2], f[u].v[3]}, ahi);
        }
      }
      Vec<V> out;
      out.v[0] = alo[0]; out.v[1] = alo[1]; out.v[2 % V] = ahi[0]; out.v[3 % V] = ahi[1];
      if (a.out_channel_major) {
#pragma unroll
        for (int v = 0; v < V; ++v) a.out_t[((size_t)b * C + c0 + v) * M + j] = out.v[v];
      } else {
        store_row<V>(a.out_t + ((size_t)b * M + j) * C + c0, out);
      }
      continue;
    }
    if (c0 >= C) continue;
    Vec<V> out;
    if constexpr (OP == OP_PSEUDOGRID) {
      constexpr int H = V == 4 ? 2 : 1;   // V == 4: two packed pairs per lane; V == 1: scalar
      constexpr int KB = 4;               // row gathers in flight per lane
      f2 wf2[kMaxKP][H];
      float wf1[kMaxKP];
#pragma unroll
      for (int p = 0; p < kMaxKP; ++p) {
        wf1[p] = 0.f;
#pragma unroll
        for (int h = 0; h < H; ++h) wf2[p][h] = pk_splat(0.f);
      }
      for (int k0 = 0; k0 < K; k0 += KB) {
        Vec<V> f[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u)
          f[u] = load_row<V>(frow + (size_t)__float_as_int(myslots[k0 + u < K ? k0 + u : K - 1].x) * C + c0);
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          if (k0 + u >= K) continue;
#pragma unroll
          for (int p4 = 0; p4 < kMaxKP / 4; ++p4) {
            const float4 h = hbuf4[p4 * TQ * KS + jq * KS + k0 + u];
            if constexpr (V == 4) {
              const f2 flo = {f[u].v[0], f[u].v[1]}, fhi = {f[u].v[2], f[u].v[3]};
              wf2[p4 * 4 + 0][0] = pk_fma(pk_splat(h.x), flo, wf2[p4 * 4 + 0][0]);
              wf2[p4 * 4 + 0][1] = pk_fma(pk_splat(h.x), fhi, wf2[p4 * 4 + 0][1]);
              wf2[p4 * 4 + 1][0] = pk_fma(pk_splat(h.y), flo, wf2[p4 * 4 + 1][0]);
              wf2[p4 * 4 + 1][1] = pk_fma(pk_splat(h.y), fhi, wf2[p4 * 4 + 1][1]);
              wf2[p4 * 4 + 2][0] = pk_fma(pk_splat(h.z), flo, wf2[p4 * 4 + 2][0]);
              wf2[p4 * 4 + 2][1] = pk_fma(pk_splat(h.z), fhi, wf2[p4 * 4 + 2][1]);
              wf2[p4 * 4 + 3][0] = pk_fma(pk_splat(h.w), flo, wf2[p4 * 4 + 3][0]);
              wf2[p4 * 4 + 3][1] = pk_fma(pk_splat(h.w), fhi, wf2[p4 * 4 + 3][1]);
            } else {
              wf1[p4 * 4 + 0] = __builtin_fmaf(h.x, f[u].v[0], wf1[p4 * 4 + 0]);
              wf1[p4 * 4 + 1] = __builtin_fmaf(h.y, f[u].v[0], wf1[p4 * 4 + 1]);
              wf1[p4 * 4 + 2] = __builtin_fmaf(h.z, f[u].v[0], wf1[p4 * 4 + 2]);
              wf1[p4 * 4 + 3] = __builtin_fmaf(h.w, f[u].v[0], wf1[p4 * 4 + 3]);
            }
          }
        }
      }
#pragma unroll
      for (int v = 0; v < V; ++v) {
        float o = 0.f;
        const int c = c0 + v;
#pragma unroll
        for (int p = 0; p < kMaxKP; ++p) {
          const float wfv = V == 4 ? wf2[p][v >> 1][v & 1] : wf1[p];
          if (p < a.pint) o += wfv * a.p1[(size_t)p * C + c];
        }
        out.v[v] = o;
      }
    } else {
      ChannelWeights<OP, V> cw;
      cw.init(a, c0);
      float acc[V];
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = 0.f;
      for_each_slot<V, 8>(myslots, K, frow, C, c0, [&](int k, const float4 &sr, const Vec<V> &f) {
        const float m = mycoef[k];
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const float t = cw.weight(v, sr.y, sr.z, sr.w) * f.v[v];  // (embedding * feature) ...
          acc[v] += t * m;                                           // ... * mask, as the reference orders it
        }
      });
#pragma unroll
      for (int v = 0; v < V; ++v) out.v[v] = a.reduction == RED_AVG ? acc[v] / n : acc[v];
    }
    if (a.out_channel_major) {  // the API layout [B,C,M], written directly instead of a transpose pass
#pragma unroll
      for (int v = 0; v < V; ++v) a.out_t[((size_t)b * C + c0 + v) * M + j] = out.v[v];
    } else {
      store_row<V>(a.out_t + ((size_t)b * M + j) * C + c0, out);  // V==4 => C%4==0: always a full vector
    }
  }
}

// ------------------------------------------------------------------------------- backward
// NP parameter-gradient accumulators per channel: adaptive 4 (3 weights + bias), pseudo grid kMaxKP.
template <int OP>
struct ParamCount {
  static constexpr int value = OP == OP_ADAPTIVE ? 4 : 0;  // PseudoGrid: d kernel_weights comes from pg_dkw_kernel
};

template <int OP, int V, bool SPARSE = false, int SBX = 4>
__global__ __launch_bounds__(256) void fused_reduce_bwd_kernel(ReduceArgs a) {
  extern __shared__ float lds[];
  // slot records staged per round; PseudoGrid also stages the kMaxKP kernel-point influences of every slot,
  // evaluated ONCE per slot by the staging threads (not once per lane group, and no cross-lane exchange)
  constexpr int kBwdCap = OP == OP_PSEUDOGRID ? 512 : 1024;
  __shared__ float4 s_rec[kBwdCap];
  __shared__ int s_qry[kBwdCap];
  constexpr bool PG_SPARSE = OP == OP_PSEUDOGRID && V == 4 && SPARSE;  // pair records from the forward pass, see pg_stage_pairs
  __shared__ float4 s_h[OP == OP_PSEUDOGRID ? kBwdCap * (PG_SPARSE ? 2 : kMaxKP / 4) : 1];
  constexpr int NP = ParamCount<OP>::value;
  const int K = a.K, C = a.C, M = a.M, N = a.N, L = a.L, QW = a.QW;
  const int waves = blockDim.x >> 6;
  const int TR = waves * QW;  // support rows per tile
  const int MK = M * K;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int g = lane / L, cl = lane - g * L;
  const bool lane_on = g < QW;
  const int tiles_per_cloud = (N + TR - 1) / TR;
  const int ntiles = a.B * tiles_per_cloud;

  for (int ch = blockIdx.y; ch < a.chunks; ch += gridDim.y) {
    const int c0 = (ch * L + cl) * V;
    const bool chan_on = lane_on && c0 < C;
    float pacc[NP > 0 ? NP : 1][V];
#pragma unroll
    for (int p = 0; p < (NP > 0 ? NP : 1); ++p)
#pragma unroll
      for (int v = 0; v < V; ++v) pacc[p][v] = 0.f;
    ChannelWeights<OP, V> cw;
    constexpr int KWR = (OP == OP_PSEUDOGRID && !PG_SPARSE) ? kMaxKP : 1;  // dense form: kernel weights in registers
    float kw[KWR][V];
#pragma unroll
    for (int p = 0; p < KWR; ++p)
#pragma unroll
      for (int v = 0; v < V; ++v) kw[p][v] = 0.f;
    if constexpr (PG_SPARSE) {  // sparse form: the chunk's kernel weights as an LDS table [kMaxKP][L*V]
      const int LV = L * V;
      __syncthreads();
      for (int t = threadIdx.x; t < kMaxKP * LV; t += (int)blockDim.x) {
        const int p = t / LV, c = ch * LV + (t - p * LV);
        lds[t] = (p < a.pint && c < C) ? a.p1[(size_t)p * C + c] : 0.f;
      }
      __syncthreads();
    }
    if (chan_on) {
      cw.init(a, c0);
      if constexpr (OP == OP_PSEUDOGRID && !PG_SPARSE) {
#pragma unroll
        for (int p = 0; p < kMaxKP; ++p)
#pragma unroll
          for (int v = 0; v < V; ++v)
            kw[p][v] = p < a.pint ? a.p1[(size_t)p * C + c0 + v] : 0.f;
      }
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      int b, tr;
      decode_tile(tile, a.B, tiles_per_cloud, b, tr);
      const int i0 = tr * TR;
      const int i = i0 + wave * QW + g;
      const bool row_on = chan_on && i < N;
      const int *off = a.inv_off + (size_t)b * (N + 1);
      const int *slots = a.inv_slots + (size_t)b * MK;
      const float4 *rec = a.slotrec + (size_t)b * MK;
      const float *grow = a.gout_t + (size_t)b * M * C + c0;
      // the slot lists of the tile's TR consecutive points are one contiguous range of inv_slots: the block
      // stages it (slot -> slotrec, query id) in LDS with every thread loading, and the lane groups walk
      // their rows out of LDS; the only global loads left in the row loop are the gout-row gathers
      const int e_lo = off[i0], e_hi = off[i0 + TR < N ? i0 + TR : N];
      const int ic = i < N ? i : N - 1;
      const int s0 = off[ic], s1 = off[ic + 1];
      Vec<V> fown;
#pragma unroll
      for (int v = 0; v < V; ++v) fown.v[v] = 0.f;
      if constexpr (NP > 0) {
        if (row_on) fown = load_row<V>(a.ft + ((size_t)b * N + i) * C + c0);
      }
      float acc[V];
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = 0.f;
      constexpr int SB = SBX;  // gout rows in flight per lane
      for (int cbeg = e_lo; cbeg < e_hi; cbeg += kBwdCap) {
        const int cn = e_hi - cbeg < kBwdCap ? e_hi - cbeg : kBwdCap;
        __syncthreads();  // the previous round's records have been consumed
        for (int t0 = 0; t0 < cn; t0 += (int)blockDim.x * 4) {
          int sl[4];
          float4 rr[4], ph[PG_SPARSE ? 4 : 1], pp[PG_SPARSE ? 4 : 1];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int t = t0 + u * (int)blockDim.x + (int)threadIdx.x;
            sl[u] = slots[cbeg + (t < cn ? t : cn - 1)];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            rr[u] = rec[sl[u]];
            if constexpr (PG_SPARSE) {
              const float4 *pr = a.pairs + ((size_t)b * MK + sl[u]) * 2;
              ph[u] = pr[0];
              pp[u] = pr[1];
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int t = t0 + u * (int)blockDim.x + (int)threadIdx.x;
            if (t < cn) {
              s_rec[t] = rr[u];
              s_qry[t] = sl[u] / K;
              if constexpr (PG_SPARSE) {
                s_h[t * 2] = ph[u];
                s_h[t * 2 + 1] = pp[u];
              } else if constexpr (OP == OP_PSEUDOGRID) {
                float h[kMaxKP];
#pragma unroll
                for (int p = 0; p < kMaxKP; ++p)
                  h[p] = p < a.pint ? kp_influence(rr[u].x, rr[u].y, rr[u].z, a.p0 + p * 3, a.pfloat, a.constant_influence) * rr[u].w
                                    : 0.f;
#pragma unroll
                for (int p4 = 0; p4 < kMaxKP / 4; ++p4)
                  s_h[t * (kMaxKP / 4) + p4] = make_float4(h[4 * p4], h[4 * p4 + 1], h[4 * p4 + 2], h[4 * p4 + 3]);
              }
            }
          }
        }
        __syncthreads();
        if (!row_on) continue;
        const int lo = s0 > cbeg ? s0 : cbeg;
        const int hi = s1 < cbeg + cn ? s1 : cbeg + cn;
      for (int e = lo; e < hi; e += SB) {
        float4 rr[SB];
        Vec<V> gg[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int t = (e + u < hi ? e + u : hi - 1) - cbeg;
          rr[u] = s_rec[t];
          gg[u] = load_row<V>(grow + (size_t)s_qry[t] * C);
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          if (e + u >= hi) continue;
          const float4 r = rr[u];
          const Vec<V> &go = gg[u];
          if constexpr (PG_SPARSE) {
            const int t = e + u - cbeg;
            f2 wlo, whi;
            pg_slot_weight(a, s_h[t * 2], *reinterpret_cast<const int4 *>(&s_h[t * 2 + 1]), lds + cl * V, L * V, r.x, r.y,
                           r.z, r.w, wlo, whi);
            acc[0] = __builtin_fmaf(wlo[0], go.v[0], acc[0]);
            acc[1 % V] = __builtin_fmaf(wlo[1], go.v[1 % V], acc[1 % V]);
            acc[2 % V] = __builtin_fmaf(whi[0], go.v[2 % V], acc[2 % V]);
            acc[3 % V] = __builtin_fmaf(whi[1], go.v[3 % V], acc[3 % V]);
          } else if constexpr (OP == OP_PSEUDOGRID) {
            float h[kMaxKP];
            {
              const int t = e + u - cbeg;
#pragma unroll
              for (int p4 = 0; p4 < kMaxKP / 4; ++p4) {
                const float4 hv = s_h[t * (kMaxKP / 4) + p4];
                h[4 * p4] = hv.x; h[4 * p4 + 1] = hv.y; h[4 * p4 + 2] = hv.z; h[4 * p4 + 3] = hv.w;
              }
            }
            if constexpr (V == 4) {
              f2 wlo = pk_splat(0.f), whi = pk_splat(0.f);
#pragma unroll
              for (int p = 0; p < kMaxKP; ++p) {
                const f2 klo = {kw[p][0], kw[p][1]}, khi = {kw[p][2], kw[p][3]};
                wlo = pk_fma(klo, pk_splat(h[p]), wlo);
                whi = pk_fma(khi, pk_splat(h[p]), whi);
              }
              acc[0] = __builtin_fmaf(wlo[0], go.v[0], acc[0]);
              acc[1] = __builtin_fmaf(wlo[1], go.v[1], acc[1]);
              acc[2] = __builtin_fmaf(whi[0], go.v[2], acc[2]);
              acc[3] = __builtin_fmaf(whi[1], go.v[3], acc[3]);
            } else {
#pragma unroll
              for (int v = 0; v < V; ++v) {
                float w = 0.f;
#pragma unroll
                for (int p = 0; p < kMaxKP; ++p) w = __builtin_fmaf(kw[p][v], h[p], w);
                acc[v] = __builtin_fmaf(w, go.v[v], acc[v]);
              }
            }
          } else {
#pragma unroll
            for (int v = 0; v < V; ++v) {
              const float gm = go.v[v] * r.w;  // d out * (mask / count)
              acc[v] = __builtin_fmaf(cw.weight(v, r.x, r.y, r.z), gm, acc[v]);
              if constexpr (OP == OP_ADAPTIVE) {
                const float gf = gm * fown.v[v];
                pacc[0][v] = __builtin_fmaf(gf, r.x, pacc[0][v]);
                pacc[1][v] = __builtin_fmaf(gf, r.y, pacc[1][v]);
                pacc[2][v] = __builtin_fmaf(gf, r.z, pacc[2][v]);
                pacc[3][v] += gf;
              }
            }
          }
        }
      }
      }  // staged rounds
      if (!row_on) continue;
      Vec<V> o;
#pragma unroll
      for (int v = 0; v < V; ++v) o.v[v] = acc[v];
      if (a.dft_channel_major) {  // the API layout [B,C,N], written directly instead of a transpose pass
#pragma unroll
        for (int v = 0; v < V; ++v) a.dft[((size_t)b * C + c0 + v) * N + i] = o.v[v];
      } else {
        store_row<V>(a.dft + ((size_t)b * N + i) * C + c0, o);
      }
    }
    // ---- fixed-order block reduction of the parameter partials for this channel chunk
    if constexpr (NP > 0) {
      if (a.dparam != nullptr) {
        const int LV = L * V;
        const int slice = LV * NP;
        __syncthreads();
        if (lane_on) {
          float *mine = lds + (size_t)(wave * QW + g) * slice + cl * V * NP;
#pragma unroll
          for (int v = 0; v < V; ++v)
#pragma unroll
            for (int p = 0; p < NP; ++p) mine[v * NP + p] = chan_on ? pacc[p][v] : 0.f;
        }
        __syncthreads();
        for (int t = threadIdx.x; t < slice; t += blockDim.x) {
          float sum = 0.f;
          for (int sl = 0; sl < waves * QW; ++sl) sum += lds[(size_t)sl * slice + t];
          const int c = ch * LV + t / NP;
          if (c < C) a.dparam[((size_t)blockIdx.x * C + c) * NP + (t - (t / NP) * NP)] = sum;
        }
        __syncthreads();
      }
    }
  }
}

// ---- PseudoGrid: d kernel_weights[p,c] = sum_{b,j} g[b,c,j] * (sum_k h_p(rel_jk) mask f[b,c,idx_jk])
// The inner sum is exactly the forward's per-query intermediate, so this is the forward loop again
// (query-major, influences prepared once per slot in LDS) with the query's output gradient folded in.
// Persistent blocks; per-lane accumulators live across tiles; one fixed-order block reduction at the end.
template <int V>
__global__ __launch_bounds__(256) void pg_dkw_kernel(ReduceArgs a) {
  extern __shared__ float4 lds4[];
  const int K = a.K, C = a.C, M = a.M, N = a.N, L = a.L, QW = a.QW;
  const int TQ = 4 * QW;
  const int KS = K + 1;  // padded per-query rows, see fused_reduce_fwd_kernel
  float4 *hbuf4 = lds4;                                    // [kMaxKP/4][TQ][KS] influences, 4 kernel points each
  int *sidx = reinterpret_cast<int *>(hbuf4 + (kMaxKP / 4) * TQ * KS);  // [TQ][KS]
  float *red = reinterpret_cast<float *>(lds4);            // reused after the tile loop
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int g = lane / L, cl = lane - g * L;
  const bool lane_on = g < QW;
  const int tiles_per_cloud = (M + TQ - 1) / TQ;
  const int ntiles = a.B * tiles_per_cloud;
  for (int ch = blockIdx.y; ch < a.chunks; ch += gridDim.y) {
    const int c0 = (ch * L + cl) * V;
    const bool chan_on = lane_on && c0 < C;
    float pacc[kMaxKP][V];
#pragma unroll
    for (int p = 0; p < kMaxKP; ++p)
#pragma unroll
      for (int v = 0; v < V; ++v) pacc[p][v] = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      int b, tq;
      decode_tile(tile, a.B, tiles_per_cloud, b, tq);
      const int j0 = tq * TQ;
      __syncthreads();
      for (int t = threadIdx.x; t < TQ * K; t += 256) {
        const int j = j0 + t / K;
        int i = 0;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < M) {
          const size_t e = ((size_t)b * M + j) * K + (t - (t / K) * K);
          i = a.idx[e];
          r = a.slotrec[e];  // {rel, mask}
        }
        const int ts = (t / K) * KS + (t - (t / K) * K);
        sidx[ts] = i;
        bool from_pairs = false;
        if constexpr (V == 4) {  // the forward pass left the slot's non-zero influences: scatter them into the dense rows
          if (a.pairs != nullptr && j < M) {
            const size_t e = ((size_t)b * M + j) * K + (t - (t / K) * K);
            const float4 hv = a.pairs[e * 2];
            const float4 pf = a.pairs[e * 2 + 1];
            const int pv[kPgPairs] = {__float_as_int(pf.x), __float_as_int(pf.y), __float_as_int(pf.z), __float_as_int(pf.w)};
            if (pv[kPgPairs - 1] >= 0) {  // (more than kPgPairs influences: evaluated below)
              from_pairs = true;
              const float hs[kPgPairs] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
              for (int p4 = 0; p4 < kMaxKP / 4; ++p4) hbuf4[p4 * TQ * KS + ts] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
              for (int q = 0; q < kPgPairs; ++q)  // unused pairs hold h = 0, p = 0 and come first in no case: written in order
                if (hs[q] != 0.f)
                  reinterpret_cast<float *>(hbuf4 + (pv[q] >> 2) * TQ * KS + ts)[pv[q] & 3] = hs[q];
            }
          }
        }
        if (!from_pairs) {
          float h[kMaxKP];
#pragma unroll
          for (int p = 0; p < kMaxKP; ++p)
            h[p] = (j < M && p < a.pint) ? kp_influence(r.x, r.y, r.z, a.p0 + p * 3, a.pfloat, a.constant_influence) * r.w : 0.f;
#pragma unroll
          for (int p4 = 0; p4 < kMaxKP / 4; ++p4)
            hbuf4[p4 * TQ * KS + ts] = make_float4(h[4 * p4], h[4 * p4 + 1], h[4 * p4 + 2], h[4 * p4 + 3]);
        }
      }
      __syncthreads();
      const int jq = wave * QW + g;
      const int j = j0 + jq;
      if (!chan_on || j >= M) continue;
      const float *frow = a.ft + (size_t)b * N * C + c0;
      // d kw[p,c] += h_p(slot) * (f[slot,c] * g[query,c]): the query's gradient is folded into the gathered
      // row first, so the slot updates the persistent accumulators directly (no per-query intermediate, 64
      // VGPRs less), and the row gathers are issued four at a time
      const Vec<V> go = load_row<V>(a.gout_t + ((size_t)b * M + j) * C + c0);
      constexpr int KB = 4;
      for (int k0 = 0; k0 < K; k0 += KB) {
        Vec<V> f[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) f[u] = load_row<V>(frow + (size_t)sidx[jq * KS + (k0 + u < K ? k0 + u : K - 1)] * C);
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          if (k0 + u >= K) continue;
          float t[V];
#pragma unroll
          for (int v = 0; v < V; ++v) t[v] = f[u].v[v] * go.v[v];
#pragma unroll
          for (int p4 = 0; p4 < kMaxKP / 4; ++p4) {
            const float4 h = hbuf4[p4 * TQ * KS + jq * KS + k0 + u];
            if constexpr (V == 4) {
              const f2 tlo = {t[0], t[1]}, thi = {t[2], t[3]};
              const float hs[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                f2 plo = {pacc[p4 * 4 + q][0], pacc[p4 * 4 + q][1]}, phi = {pacc[p4 * 4 + q][2], pacc[p4 * 4 + q][3]};
                plo = pk_fma(pk_splat(hs[q]), tlo, plo);
                phi = pk_fma(pk_splat(hs[q]), thi, phi);
                pacc[p4 * 4 + q][0] = plo[0]; pacc[p4 * 4 + q][1] = plo[1];
                pacc[p4 * 4 + q][2] = phi[0]; pacc[p4 * 4 + q][3] = phi[1];
              }
            } else {
#pragma unroll
              for (int v = 0; v < V; ++v) {
                pacc[p4 * 4 + 0][v] = __builtin_fmaf(h.x, t[v], pacc[p4 * 4 + 0][v]);
                pacc[p4 * 4 + 1][v] = __builtin_fmaf(h.y, t[v], pacc[p4 * 4 + 1][v]);
                pacc[p4 * 4 + 2][v] = __builtin_fmaf(h.z, t[v], pacc[p4 * 4 + 2][v]);
                pacc[p4 * 4 + 3][v] = __builtin_fmaf(h.w, t[v], pacc[p4 * 4 + 3][v]);
              }
            }
          }
        }
      }
    }
    // fixed-order block reduction, eight kernel points at a time (keeps the LDS slices at 32 KiB)
    const int LV = L * V;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int slice = LV * 8;
      __syncthreads();
      if (lane_on) {
        float *mine = red + (size_t)(wave * QW + g) * slice + cl * V * 8;
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
          for (int p = 0; p < 8; ++p) mine[v * 8 + p] = chan_on ? pacc[half * 8 + p][v] : 0.f;
      }
      __syncthreads();
      for (int t = threadIdx.x; t < slice; t += 256) {
        float sum = 0.f;
        for (int sl = 0; sl < 4 * QW; ++sl) sum += red[(size_t)sl * slice + t];
        const int c = ch * LV + t / 8;
        if (c < C) a.dparam[((size_t)blockIdx.x * C + c) * kMaxKP + half * 8 + (t & 7)] = sum;
      }
    }
    __syncthreads();
  }
}

static int check_common(const ReduceArgs &a, const char *who) {
  if (a.B < 0 || a.N < 1 || a.M < 0 || a.K < 1 || a.C < 1) return fail(CL3D_E_INVALID, "%s: bad sizes", who);
  if ((long long)a.M * a.K > 0x7fffffffLL) return fail(CL3D_E_UNSUPPORTED, "%s: M*K too large", who);
  return CL3D_OK;
}

static int validate_op(int op, int C, int pint, const char *who) {
  switch (op) {
    case OP_POSPOOL_XYZ:
      if (C % 3) return fail(CL3D_E_INVALID, "%s: PosPool xyz needs C %% 3 == 0 (C=%d)", who, C);
      return CL3D_OK;
    case OP_POSPOOL_SINCOS:
      if (C % 6) return fail(CL3D_E_INVALID, "%s: PosPool sin_cos needs C %% 6 == 0 (C=%d)", who, C);
      return CL3D_OK;
    case OP_ADAPTIVE:
      if (pint < 1 || C % pint) return fail(CL3D_E_INVALID, "%s: shared_channels=%d does not divide C=%d", who, pint, C);
      return CL3D_OK;
    case OP_PSEUDOGRID:
      if (pint < 1 || pint > kMaxKP) return fail(CL3D_E_UNSUPPORTED, "%s: %d kernel points (max %d)", who, pint, kMaxKP);
      return CL3D_OK;
    default:
      return fail(CL3D_E_INVALID, "%s: unknown operator %d", who, op);
  }
}

// PseudoGrid takes the sparse (pair record) form when the lanes hold four channels and the influence is not constant
static bool pg_sparse(int C, int constant_influence) { return C % 4 == 0 && !constant_influence; }

template <int V>
static void launch_fwd(int op, const ReduceArgs &a, dim3 grid, size_t lds, hipStream_t st) {
  switch (op) {
    case OP_POSPOOL_XYZ: hipLaunchKernelGGL((fused_reduce_fwd_kernel<OP_POSPOOL_XYZ, V>), grid, dim3(256), lds, st, a); break;
    case OP_POSPOOL_SINCOS: hipLaunchKernelGGL((fused_reduce_fwd_kernel<OP_POSPOOL_SINCOS, V>), grid, dim3(256), lds, st, a); break;
    case OP_ADAPTIVE: hipLaunchKernelGGL((fused_reduce_fwd_kernel<OP_ADAPTIVE, V>), grid, dim3(256), lds, st, a); break;
    default:
      if (pg_sparse(a.C, a.constant_influence)) hipLaunchKernelGGL((fused_reduce_fwd_kernel<OP_PSEUDOGRID, V, true>), grid, dim3(256), lds, st, a);
      else hipLaunchKernelGGL((fused_reduce_fwd_kernel<OP_PSEUDOGRID, V>), grid, dim3(256), lds, st, a);
      break;
  }
}

template <int V>
static void launch_bwd(int op, const ReduceArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  switch (op) {
    // eight gout rows in flight per lane for the cheapest weight (58 -> ~90 registers, still 5 waves / SIMD): PosPool
    // step 0.288 -> 0.285 ms; AdaptiveWeight, which carries its parameter partials, loses 4 % that way and keeps four
    case OP_POSPOOL_XYZ: hipLaunchKernelGGL((fused_reduce_bwd_kernel<OP_POSPOOL_XYZ, V, false, 8>), grid, block, lds, st, a); break;
    case OP_POSPOOL_SINCOS: hipLaunchKernelGGL((fused_reduce_bwd_kernel<OP_POSPOOL_SINCOS, V>), grid, block, lds, st, a); break;
    case OP_ADAPTIVE: hipLaunchKernelGGL((fused_reduce_bwd_kernel<OP_ADAPTIVE, V>), grid, block, lds, st, a); break;
    default:
      if (pg_sparse(a.C, a.constant_influence)) hipLaunchKernelGGL((fused_reduce_bwd_kernel<OP_PSEUDOGRID, V, true>), grid, block, lds, st, a);
      else hipLaunchKernelGGL((fused_reduce_bwd_kernel<OP_PSEUDOGRID, V>), grid, block, lds, st, a);
      break;
  }
}

static LaneMap fwd_lane_map(int op, int C, int K, int V, size_t *lds_out, bool sparse = false) {
  LaneMap m = pick_lane_map(C, V, 64, op != OP_POSPOOL_SINCOS);
  if (m.QW > 16) {  // keep the per-block slot tile modest
    m.QW = 16;
    m.L = 4;
    m.chunks = ((C + V - 1) / V + m.L - 1) / m.L;
  }
  for (;;) {
    const size_t tq = 4 * (size_t)m.QW;
    size_t lds = tq * (K + 1) * (sizeof(float4) + sizeof(float)) + tq * sizeof(float);
    if (op == OP_PSEUDOGRID)  // sparse: pair records + the chunk's kernel-weight table; dense: all influences
      lds += sparse ? tq * (K + 1) * 2 * sizeof(float4) + (size_t)kMaxKP * m.L * V * sizeof(float) : tq * (K + 1) * kMaxKP * sizeof(float);
    if (lds <= 60 * 1024 || m.QW == 1) {
      *lds_out = lds;
      return m;
    }
    m.QW -= 1;  // fewer queries per wave (some lanes idle) until the tile fits
  }
}

// the d kernel_weights pass keeps dense influences of its tile and reduces 8 kernel points at a time through LDS
static LaneMap dkw_lane_map(int C, int K, int V, size_t *lds_out) {
  size_t unused = 0;
  LaneMap m = fwd_lane_map(OP_PSEUDOGRID, C, K, V, &unused, false);
  for (;;) {
    const size_t tile = 4 * (size_t)m.QW * (K + 1) * (sizeof(int) + kMaxKP * sizeof(float));
    const size_t red = 4 * (size_t)m.QW * m.L * V * 8 * sizeof(float);
    const size_t lds = tile > red ? tile : red;
    if (lds <= 60 * 1024 || m.QW == 1) {
      *lds_out = lds;
      return m;
    }
    m.QW -= 1;
  }
}

bool fused_reduce_supported(int op, int K, int C) {
  if (op < OP_POSPOOL_XYZ || op > OP_PSEUDOGRID || K < 1 || C < 1) return false;
  const int V = (C % 4 == 0) ? 4 : 1;
  size_t lds = 0;
  fwd_lane_map(op, C, K, V, &lds, false);
  if (lds > 64 * 1024) return false;
  if (op == OP_PSEUDOGRID) {
    if (V == 4) {
      fwd_lane_map(op, C, K, V, &lds, true);
      if (lds > 64 * 1024) return false;
    }
    dkw_lane_map(C, K, V, &lds);
    if (lds > 64 * 1024) return false;
  }
  return true;
}

// ---- the blocks' parameter-gradient partials dparam [G, C, NP] summed in double in a fixed order, written in the
// parameters' own layouts: PseudoGrid d kernel_weights [P, C]; AdaptiveWeight (S consecutive channels share one weight
// row) d W [C/S, 3] and d bias [C/S].  One workgroup per output row (a channel / a group of S channels); thread =
// (partial-block lane, column), eight loads in flight.  Replaces five to six library launches per step (cast to double,
// sum, slice, transpose, cast back: 33 us of the replayed PseudoGrid step) with one.
template <int NP>
__global__ __launch_bounds__(256) void param_reduce_kernel(const float *__restrict__ dparam, int G, int C, int S,
                                                           int pint, int adaptive, float *__restrict__ out0,
                                                           float *__restrict__ out1) {
  __shared__ double s_red[256];
  constexpr int LG = 256 / NP, UB = 8;
  const int row = blockIdx.x;
  const int k = threadIdx.x % NP, gl = threadIdx.x / NP;
  double acc = 0.0;
  for (int s = 0; s < S; ++s) {
    const float *col = dparam + (size_t)(row * S + s) * NP + k;
    for (int g0 = gl; g0 < G; g0 += LG * UB) {
      float v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int g = g0 + u * LG;
        v[u] = col[(size_t)(g < G ? g : G - 1) * C * NP];
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (g0 + u * LG < G) acc += (double)v[u];
    }
  }
  s_red[threadIdx.x] = acc;
  __syncthreads();
  for (int stride = LG / 2; stride >= 1; stride >>= 1) {  // the partial-block lanes folded pairwise, always the same pairs
    if (gl < stride) s_red[threadIdx.x] += s_red[threadIdx.x + stride * NP];
    __syncthreads();
  }
  if (gl != 0) return;
  const float r = (float)s_red[k];
  if (adaptive) {
    if (k < 3) out0[row * 3 + k] = r;
    else if (k == 3) out1[row] = r;
  } else if (k < pint) {
    out1[(size_t)k * C + row] = r;
  }
}

}  // namespace cl3d

extern "C" int cl3d_fused_param_reduce(int op, const float *dparam, int n_partials, int C, int pint, float *g0,
                                       float *g1, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(op == OP_ADAPTIVE || op == OP_PSEUDOGRID, "fused_param_reduce: operator without parameters");
  CL3D_REQUIRE(dparam && g1 && n_partials >= 1 && C >= 1 && pint >= 1, "fused_param_reduce: bad arguments");
  if (op == OP_ADAPTIVE) {
    CL3D_REQUIRE(g0 && C % pint == 0, "fused_param_reduce: AdaptiveWeight needs d W and C %% shared_channels == 0 (C=%d, shared_channels=%d)", C, pint);
    hipLaunchKernelGGL(param_reduce_kernel<4>, dim3(C / pint), dim3(256), 0, (hipStream_t)stream, dparam, n_partials, C,
                       pint, pint, 1, g0, g1);
  } else {
    CL3D_REQUIRE(pint <= kMaxKP, "fused_param_reduce: too many kernel points");
    hipLaunchKernelGGL(param_reduce_kernel<kMaxKP>, dim3(C), dim3(256), 0, (hipStream_t)stream, dparam, n_partials, C,
                       1, pint, 0, g0, g1);
  }
  return check_launch("cl3d_fused_param_reduce");
}

extern "C" int cl3d_fused_param_partials(int op, int B, int N, int C) {
  (void)C;
  if (op != cl3d::OP_ADAPTIVE && op != cl3d::OP_PSEUDOGRID) return 0;
  return cl3d::round_grid(((long long)B * N + 3) / 4, 1024);
}

extern "C" int cl3d_fused_reduce_fwd(int op, const float *query_xyz, const float *support_xyz,
                                     const int32_t *query_mask, const int32_t *idx,
                                     const int32_t *idx_mask, const float *ft, int B, int N, int M,
                                     int K, int C, float radius, int normalize_xyz, int reduction,
                                     const float *p0, const float *p1, int pint, float pfloat,
                                     int constant_influence, float *out, int out_channel_major,
                                     float *slotrec, float *pairs, cl3d_stream_t stream) {
  using namespace cl3d;
  ReduceArgs a{};
  a.query_xyz = query_xyz; a.support_xyz = support_xyz; a.query_mask = query_mask; a.idx = idx; a.idx_mask = idx_mask;
  a.ft = ft; a.p0 = p0; a.p1 = p1; a.out_t = out; a.out_channel_major = out_channel_major;
  a.slotrec = reinterpret_cast<float4 *>(slotrec);
  a.pairs = (op == OP_PSEUDOGRID && pg_sparse(C, constant_influence)) ? reinterpret_cast<float4 *>(pairs) : nullptr;
  a.B = B; a.N = N; a.M = M; a.K = K; a.C = C;
  a.reduction = reduction; a.normalize = normalize_xyz; a.pint = pint; a.constant_influence = constant_influence;
  a.inv_radius = 1.0f / radius; a.pfloat = pfloat;
  int rc = check_common(a, "fused_reduce_fwd");
  if (rc != CL3D_OK) return rc;
  rc = validate_op(op, C, pint, "fused_reduce_fwd");
  if (rc != CL3D_OK) return rc;
  CL3D_REQUIRE(reduction == RED_SUM || reduction == RED_AVG, "fused_reduce_fwd: reduction must be sum or avg");
  if (B == 0 || M == 0) return CL3D_OK;
  CL3D_REQUIRE(query_xyz && support_xyz && query_mask && idx && idx_mask && ft && out, "fused_reduce_fwd: null pointer");
  CL3D_REQUIRE(op == OP_POSPOOL_XYZ || p0, "fused_reduce_fwd: missing operator parameters");
  const int V = (C % 4 == 0) ? 4 : 1;
  size_t lds = 0;
  const LaneMap m = fwd_lane_map(op, C, K, V, &lds, op == OP_PSEUDOGRID && pg_sparse(C, constant_influence));
  if (lds > 64 * 1024) return fail(CL3D_E_UNSUPPORTED, "fused_reduce_fwd: nsample=%d needs %zu B of LDS per block", K, lds);
  a.L = m.L; a.QW = m.QW; a.chunks = m.chunks;
  const int tiles_fwd = virtual_tiles(B, ceil_div(M, 4 * m.QW));
  dim3 grid(tiles_fwd, chunk_grid(tiles_fwd, m.chunks));
  if (V == 4) launch_fwd<4>(op, a, grid, lds, (hipStream_t)stream);
  else launch_fwd<1>(op, a, grid, lds, (hipStream_t)stream);
  return check_launch("cl3d_fused_reduce_fwd");
}

extern "C" int cl3d_fused_reduce_bwd(int op, const float *gout_t, const float *ft,
                                     const float *slotrec, const float *pairs, const int32_t *idx, const int32_t *inv_off,
                                     const int32_t *inv_slots, int B, int N, int M, int K, int C,
                                     const float *p0, const float *p1, int pint, float pfloat,
                                     int constant_influence, float *dft, int dft_channel_major, float *dparam,
                                     int n_partials, cl3d_stream_t stream) {
  using namespace cl3d;
  ReduceArgs a{};
  a.gout_t = gout_t; a.ft = ft; a.slotrec = reinterpret_cast<float4 *>(const_cast<float *>(slotrec));
  a.pairs = reinterpret_cast<float4 *>(const_cast<float *>(pairs));
  if (op != OP_PSEUDOGRID || !pg_sparse(C, constant_influence)) a.pairs = nullptr;  // only the sparse forward writes them
  a.idx = idx; a.inv_off = inv_off; a.inv_slots = inv_slots; a.p0 = p0; a.p1 = p1; a.dft = dft; a.dft_channel_major = dft_channel_major; a.dparam = dparam;
  a.B = B; a.N = N; a.M = M; a.K = K; a.C = C; a.pint = pint; a.pfloat = pfloat; a.constant_influence = constant_influence;
  int rc = check_common(a, "fused_reduce_bwd");
  if (rc != CL3D_OK) return rc;
  rc = validate_op(op, C, pint, "fused_reduce_bwd");
  if (rc != CL3D_OK) return rc;
  if (B == 0) return CL3D_OK;
  CL3D_REQUIRE(gout_t && slotrec && inv_off && inv_slots && dft, "fused_reduce_bwd: null pointer");
  CL3D_REQUIRE(op == OP_POSPOOL_XYZ || p0, "fused_reduce_bwd: missing operator parameters");
  CL3D_REQUIRE(op != OP_PSEUDOGRID || !pg_sparse(C, constant_influence) || pairs,
               "fused_reduce_bwd: PseudoGrid needs the forward pass's pair records");
  const bool has_params = op == OP_ADAPTIVE || op == OP_PSEUDOGRID;
  CL3D_REQUIRE(!has_params || (ft && dparam && n_partials == cl3d_fused_param_partials(op, B, N, C)),
               "fused_reduce_bwd: parameter-gradient buffer must have cl3d_fused_param_partials() blocks");
  const int V = (C % 4 == 0) ? 4 : 1;
  LaneMap m = pick_lane_map(C, V, 64, op != OP_POSPOOL_SINCOS);
  // PseudoGrid carries 2*kMaxKP*V accumulators per lane: two waves per block keep the LDS slice at 32 KiB
  const int waves = 4;
  const int NP = op == OP_ADAPTIVE ? 4 : 0;  // PseudoGrid's parameter gradient has its own kernel below
  size_t lds = (size_t)waves * m.QW * m.L * V * NP * sizeof(float);
  if (op == OP_PSEUDOGRID && pg_sparse(C, constant_influence)) lds = (size_t)kMaxKP * m.L * V * sizeof(float);  // kernel-weight table of a channel chunk
  a.L = m.L; a.QW = m.QW; a.chunks = m.chunks;
  const long long tiles = (long long)B * ceil_div(N, waves * m.QW);
  const int gx = has_params ? n_partials : round_grid(tiles, 4096);
  const int gy = chunk_grid(tiles < gx ? tiles : gx, m.chunks);
  if (V == 4) launch_bwd<4>(op, a, dim3(gx, gy), dim3(64 * waves), lds, (hipStream_t)stream);
  else launch_bwd<1>(op, a, dim3(gx, gy), dim3(64 * waves), lds, (hipStream_t)stream);
  rc = check_launch("cl3d_fused_reduce_bwd");
  if (rc != CL3D_OK || op != OP_PSEUDOGRID) return rc;
  // d kernel_weights: query-major pass (the forward loop with the output gradient folded in)
  CL3D_REQUIRE(idx, "fused_reduce_bwd: PseudoGrid needs idx");
  size_t lds_dkw = 0;
  const LaneMap mf = dkw_lane_map(C, K, V, &lds_dkw);
  a.L = mf.L; a.QW = mf.QW; a.chunks = mf.chunks;
  if (lds_dkw > 64 * 1024) return fail(CL3D_E_UNSUPPORTED, "fused_reduce_bwd: nsample=%d needs %zu B of LDS", K, lds_dkw);
  const long long tiles_q = (long long)B * ceil_div(M, 4 * mf.QW);
  const dim3 grid_dkw(n_partials, chunk_grid(tiles_q < n_partials ? tiles_q : n_partials, mf.chunks));
  // (round 6: this pass forked onto a stream of the library's, beside the support-major pass instead of behind it, was
  // built and measured again -- scripts/micro/kernel_variants.py "pg_nofork", profiles/r06/session7_summary.txt: side by
  // side the two take 167 + 105 us instead of 83 + 96 one behind the other -- both are bound by issue and L2 requests, not
  // by exposed latency -- and the step 0.389-0.390 against 0.384-0.386 ms; as in round 3, dropped)
  // (a sparse form of this pass -- per-lane-group accumulators in LDS, updated per pair -- was built and measured: 188 us
  // against 95 us for the dense loop, whose 32 packed FMAs per slot cost no more than the ballots, address arithmetic
  // and dependent LDS read-modify-writes of ~1.2 updates; the gradient of the kernel weights stays dense)
  if (V == 4) hipLaunchKernelGGL((pg_dkw_kernel<4>), grid_dkw, dim3(256), lds_dkw, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((pg_dkw_kernel<1>), grid_dkw, dim3(256), lds_dkw, (hipStream_t)stream, a);
  return check_launch("cl3d_fused_reduce_bwd(dkw)");
}

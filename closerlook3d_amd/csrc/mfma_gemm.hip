// mfma_gemm.hip -- the dense per-point contractions of the hot path on the gfx950 matrix cores.
//
// north_star: "MFMA used only for the dense (B*N*K, C_in) x (C_in, C_out) neighbourhood-feature contraction".
// The engine factors that contraction (reference local_aggregation_operators.py:253-257,288-295: Conv2d 3+2C -> Co
// over all B*M*K neighbourhood positions) into  y = W_r rel + H[centre] + G[neighbour]  with one row
// [G_i | H_i] = [W_d f_i | (W_c - W_d) f_i]  per POINT (fused_pwmlp.hip), so what is left of it is a per-point GEMM
// [B*N, C] x [C, 2Co] -- K = nsample times fewer flops -- and its two gradients.  The same kernel serves the 1x1
// Conv1d layers either side of the operator (reference backbones/resnet.py:32-39,58-66: conv1 / conv2 / shortcut),
// which are per-point GEMMs on channel-major tensors.
//
// One kernel template:   D[i][j] = sum_k A(i,k) * B(j,k)
//   * both operand tiles are staged through LDS as T[k][r] (r = i or j contiguous) whatever their layout in HBM:
//     a source whose r axis is contiguous (channel-major features: r = point) is copied with 16-byte loads and
//     16-byte LDS stores; a source whose k axis is contiguous (point-major rows, weight rows) is read with 16-byte
//     loads along k -- full 128-byte lines per row -- and transposed by the LDS write (row stride odd => the four
//     scalar stores of a lane group hit 32 distinct banks).  So the layout change channel-major <-> point-major at
//     the operator boundary costs nothing: it is the direction in which a tile is written to / read from LDS;
//   * MFMA fragments then are single conflict-free LDS reads: lane l holds T[k0 + (l>>5)][r0 + (l&31)] for
//     v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain), or the 16-byte bf16 pack
//     T[(k0>>3) + (l>>5)][r0 + (l&31)][0..8) for v_mfma_f32_32x32x16_bf16 (inputs rounded to bf16 -- RNE,
//     v_cvt_pk_bf16_f32 -- while they are staged; accumulation and everything outside the contraction stay f32);
//   * which operand plays "A" decides the register layout of D (a lane holds one column j, 16 rows i), so the side
//     whose index is contiguous in the OUTPUT is always put on j: every store instruction writes 128-byte runs;
//   * a 256-thread workgroup = 2 x 2 waves, each wave a WI x WJ grid of 32x32 accumulators (workgroup tiles 128x128,
//     128x64, 64x128 or 64x64, picked per shape so that the chip is filled and little of a tile is padding); the next
//     K chunk is prefetched into registers while the current one is multiplied out of LDS;
//   * clouds are folded into the point axis: a tile of 128 points may span several small clouds of a channel-major
//     tensor (the deep stages have 16 clouds x 16..64 points), index -> (cloud, point) per load;
//   * weight gradients contract over ALL points (K = B*N): the (batch, point) axis is cut into contiguous slices,
//     one workgroup each, partial products go to scratch and are summed in slice order by a second small kernel
//     (no atomics: bit-reproducible), which also folds the PointWiseMLP weight plumbing
//     (d W_c = bot, d W_d = top - bot, d W_r) so no separate merge pass exists on this path.
// The PointWiseMLP weight [Co, 3+2C] = [W_r | W_c | W_d] is turned into wcat = [W_d ; W_c - W_d] and W_r by one small
// launch ahead of the forward GEMM (pwmlp_weights_kernel); the weight-gradient reduce writes d W directly.
#include "cl3d_common.h"
#include <stdlib.h>

namespace cl3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { PREC_F32 = 0, PREC_BF16 = 1 };
enum { STAGE_VEC_RC = 0, STAGE_VEC_KC = 1, STAGE_SCALAR = 2 };  // how an operand tile travels HBM -> LDS

struct GemmOperand {
  const float *p;
  int sr, sk;            // element strides along the tile index r and the contraction index k
  int R;                 // extent along r (clouds folded in)
  int rc;                // 1: r is the contiguous axis (sr == 1), 0: k is (sk == 1)
  int vec;               // 16-byte loads along the contiguous axis are legal (alignment and extents)
  int fold;              // 0: none; 1: r = cloud * fold_n + point; 2: k = cloud * fold_n + point  (channel-major tensors)
  int fold_n;            // points per cloud
  long long sb;          // cloud stride of a folded axis
};

struct GemmArgs {
  GemmOperand A, B;  // D[i][j] = sum_k A(i,k) B(j,k)
  float *D;
  long long d_si, d_sj;
  int d_fold_n;        // != 0: j = cloud * d_fold_n + point, cloud stride d_sb (channel-major output)
  long long d_sb;
  int K;               // contraction extent (clouds folded in for weight gradients)
  int nsplit, chunks_per_split;  // nsplit > 1: K cut into slices, slice s writes its partial tile to D + s*I*J
  int tiles_i, tiles_j;
  // inference epilogue (BatchNorm folded to a per-row affine map, residual, ReLU): D = act(scale[i] * acc + shift[i] + res)
  const float *ep_scale, *ep_shift, *ep_res;
  int ep_relu;
};

// element offset of (r, k) of an operand; the pointer base is wave-uniform, the offset a 32-bit lane value
__device__ __forceinline__ unsigned gemm_off(const GemmOperand &s, int r, int k) {
  if (s.fold == 1) {
    const int b = r / s.fold_n;
    return (unsigned)(b * (int)s.sb + (r - b * s.fold_n) * s.sr + k * s.sk);
  }
  if (s.fold == 2) {
    const int b = k / s.fold_n;
    return (unsigned)(b * (int)s.sb + r * s.sr + (k - b * s.fold_n) * s.sk);
  }
  return (unsigned)(r * s.sr + k * s.sk);
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float &comp(float4 &v, int e) { return reinterpret_cast<float *>(&v)[e]; }

// ---- staging, f32: LDS tile T[KC][TR + 4] floats (row stride TR+4 for an r-contiguous source, TR+1 for a
// k-contiguous one: odd, so the transposing scalar stores are conflict-free) ------------------------------------
template <int TR, int KC>
struct StageF32 {
  static constexpr int NV = TR * KC / 1024;  // float4 per thread and chunk
  static constexpr int kLdsFloats = KC * (TR + 4);
  float4 v[NV];

  __device__ __forceinline__ static int stride(const GemmOperand &s) { return s.rc ? TR + 4 : TR + 1; }

  template <int MODE>
  __device__ __forceinline__ void load(const GemmOperand &s, int r0, int k0, int K) {
    const int t = threadIdx.x;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == STAGE_VEC_RC) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int idx = q * 256 + t, r = r0 + 4 * (idx % (TR / 4)), k = k0 + idx / (TR / 4);
        v[q] = (r < s.R && k < K) ? ld4(s.p + gemm_off(s, r, k)) : zero;
      }
    } else if (MODE == STAGE_VEC_KC) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int idx = q * 256 + t, k = k0 + 4 * (idx % (KC / 4)), r = r0 + idx / (KC / 4);
        v[q] = (r < s.R && k < K) ? ld4(s.p + gemm_off(s, r, k)) : zero;
      }
    } else {
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = (q * 4 + e) * 256 + t;
          const int r = r0 + (s.rc ? idx % TR : idx / KC), k = k0 + (s.rc ? idx / TR : idx % KC);
          comp(v[q], e) = (r < s.R && k < K) ? s.p[gemm_off(s, r, k)] : 0.f;
        }
    }
  }

  template <int MODE>
  __device__ __forceinline__ void store(const GemmOperand &s, float *T) const {
    const int t = threadIdx.x;
    if (MODE == STAGE_VEC_RC) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int idx = q * 256 + t;
        *reinterpret_cast<float4 *>(T + (idx / (TR / 4)) * (TR + 4) + 4 * (idx % (TR / 4))) = v[q];
      }
    } else if (MODE == STAGE_VEC_KC) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int idx = q * 256 + t, k = 4 * (idx % (KC / 4)), r = idx / (KC / 4);
        float4 x = v[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) T[(k + e) * (TR + 1) + r] = comp(x, e);
      }
    } else {
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = (q * 4 + e) * 256 + t;
          float4 x = v[q];
          if (s.rc) T[(idx / TR) * (TR + 4) + idx % TR] = comp(x, e);
          else T[(idx % KC) * (TR + 1) + idx / KC] = comp(x, e);
        }
    }
  }
};

// ---- staging, bf16: LDS tile T[KC/8][TR + 1] packs of 8 bf16 (16 bytes) along k; KC = 64 --------------------
template <int TR>
struct StageBF16 {
  static constexpr int KC = 64, G = KC / 8;
  static constexpr int kLdsPacks = G * (TR + 1);
  static constexpr int NQ = (TR * G + 255) / 256;  // (r, g) items per thread on the k-contiguous / scalar paths
  float4 v[8];

  template <int MODE>
  __device__ __forceinline__ void load(const GemmOperand &s, int r0, int k0, int K) {
    const int t = threadIdx.x;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == STAGE_VEC_RC) {  // one (k group, 4 rows) item per thread: 8 loads, each coalesced over the wave
      const int g = t / (TR / 4), r = r0 + 4 * (t % (TR / 4));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + 8 * g + e;
        v[e] = (g < G && r < s.R && k < K) ? ld4(s.p + gemm_off(s, r, k)) : zero;
      }
    } else if (MODE == STAGE_VEC_KC) {  // (row, k group) items: 32 contiguous bytes, 8 lanes cover 256 bytes of a row
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int idx = q * 256 + t, g = idx % G, r = r0 + idx / G, k = k0 + 8 * g;
        const bool ok = idx < TR * G && r < s.R;
        v[2 * q] = (ok && k < K) ? ld4(s.p + gemm_off(s, r, k)) : zero;
        v[2 * q + 1] = (ok && k + 4 < K) ? ld4(s.p + gemm_off(s, r, k + 4)) : zero;
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int idx = q * 256 + t;
        const int g = s.rc ? idx / TR : idx % G, r = r0 + (s.rc ? idx % TR : idx / G);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = k0 + 8 * g + e;
          comp(v[2 * q + e / 4], e % 4) = (idx < TR * G && r < s.R && k < K) ? s.p[gemm_off(s, r, k)] : 0.f;
        }
      }
    }
  }

  __device__ __forceinline__ static uint4 pack(const float *x) {
    bf16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (__bf16)x[e];  // v_cvt_pk_bf16_f32: round to nearest even
    return *reinterpret_cast<uint4 *>(&h);
  }

  template <int MODE>
  __device__ __forceinline__ void store(const GemmOperand &s, uint4 *T) const {
    const int t = threadIdx.x;
    if (MODE == STAGE_VEC_RC) {
      const int g = t / (TR / 4), r = 4 * (t % (TR / 4));
      if (g < G) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float4 w = v[e];
            x[e] = comp(w, i);
          }
          T[g * (TR + 1) + r + i] = pack(x);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int idx = q * 256 + t;
        const bool gfast = MODE == STAGE_VEC_KC || !s.rc;
        const int g = gfast ? idx % G : idx / TR, r = gfast ? idx / G : idx % TR;
        float x[8];
        float4 lo = v[2 * q], hi = v[2 * q + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x[e] = comp(lo, e);
          x[4 + e] = comp(hi, e);
        }
        if (idx < TR * G) T[g * (TR + 1) + r] = pack(x);
      }
    }
  }
};

template <int PREC, int TR>
struct StagePick {
  using type = StageF32<TR, 32>;
  static constexpr int KC = 32;
  static constexpr int kLdsBytes = StageF32<TR, 32>::kLdsFloats * 4;
};
template <int TR>
struct StagePick<PREC_BF16, TR> {
  using type = StageBF16<TR>;
  static constexpr int KC = 64;
  static constexpr int kLdsBytes = StageBF16<TR>::kLdsPacks * 16;
};

template <int PREC, int WI, int WJ, int AM, int BM>
__global__ __launch_bounds__(256, 2) void mfma_gemm_kernel(GemmArgs a) {
  constexpr int TI = 64 * WI, TJ = 64 * WJ;
  using SA = StagePick<PREC, TI>;
  using SB = StagePick<PREC, TJ>;
  constexpr int KC = SA::KC;
  __shared__ __attribute__((aligned(16))) unsigned char lds[SA::kLdsBytes + SB::kLdsBytes];
  unsigned char *ldsA = lds, *ldsB = lds + SA::kLdsBytes;

  // tile / K slice of this workgroup
  int bid = blockIdx.x;
  const int tj = bid % a.tiles_j;
  bid /= a.tiles_j;
  const int ti = bid % a.tiles_i;
  const int z = bid / a.tiles_i;  // K slice
  const int i0 = ti * TI, j0 = tj * TJ;
  const int chunks = (a.K + KC - 1) / KC;
  int g0 = 0, g1 = chunks;
  if (a.nsplit > 1) {
    g0 = z * a.chunks_per_split;
    g1 = g0 + a.chunks_per_split;
    if (g1 > chunks) g1 = chunks;
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wi0 = (wave >> 1) * 32 * WI, wj0 = (wave & 1) * 32 * WJ;
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[WI][WJ];
#pragma unroll
  for (int x = 0; x < WI; ++x)
#pragma unroll
    for (int y = 0; y < WJ; ++y)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[x][y][e] = 0.f;

  // Two K chunks are kept in flight in registers (sets 0 and 1): a chunk is written to LDS two loop turns after its
  // loads were issued, so the write never waits on HBM latency even with one wave per SIMD.
  // (the widest register footprints -- bf16 at 128 x 128, the scalar fallback -- keep a single set)
  constexpr int DEPTH = ((PREC == PREC_BF16 && WI * WJ == 4) || AM == STAGE_SCALAR) ? 1 : 2;
  typename SA::type sa[DEPTH];
  typename SB::type sb[DEPTH];
  auto issue = [&](int set, int g) {
    if (set == 0 || DEPTH == 1) {
      sa[0].template load<AM>(a.A, i0, g * KC, a.K);
      sb[0].template load<BM>(a.B, j0, g * KC, a.K);
    } else {
      sa[DEPTH - 1].template load<AM>(a.A, i0, g * KC, a.K);
      sb[DEPTH - 1].template load<BM>(a.B, j0, g * KC, a.K);
    }
  };
  auto multiply = [&]() {
    if constexpr (PREC == PREC_F32) {
      const float *TA = reinterpret_cast<const float *>(ldsA), *TB = reinterpret_cast<const float *>(ldsB);
      const int strA = SA::type::stride(a.A), strB = SB::type::stride(a.B);
#pragma unroll 4
      for (int kk = 0; kk < KC / 2; ++kk) {
        float fa[WI], fb[WJ];
#pragma unroll
        for (int x = 0; x < WI; ++x) fa[x] = TA[(2 * kk + lh) * strA + wi0 + 32 * x + lr];
#pragma unroll
        for (int y = 0; y < WJ; ++y) fb[y] = TB[(2 * kk + lh) * strB + wj0 + 32 * y + lr];
#pragma unroll
        for (int x = 0; x < WI; ++x)
#pragma unroll
          for (int y = 0; y < WJ; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[x], fb[y], acc[x][y], 0, 0, 0);
      }
    } else {
      const uint4 *TA = reinterpret_cast<const uint4 *>(ldsA), *TB = reinterpret_cast<const uint4 *>(ldsB);
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) {
        uint4 fa[WI], fb[WJ];
#pragma unroll
        for (int x = 0; x < WI; ++x) fa[x] = TA[(2 * ks + lh) * (TI + 1) + wi0 + 32 * x + lr];
#pragma unroll
        for (int y = 0; y < WJ; ++y) fb[y] = TB[(2 * ks + lh) * (TJ + 1) + wj0 + 32 * y + lr];
#pragma unroll
        for (int x = 0; x < WI; ++x)
#pragma unroll
          for (int y = 0; y < WJ; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8 *>(&fa[x]),
                                                                *reinterpret_cast<bf16x8 *>(&fb[y]), acc[x][y], 0, 0, 0);
      }
    }
  };
  auto to_lds = [&](int set) {
    __syncthreads();  // the previous chunk has been multiplied out of LDS
    if constexpr (PREC == PREC_F32) {
      if (set == 0 || DEPTH == 1) {
        sa[0].template store<AM>(a.A, reinterpret_cast<float *>(ldsA));
        sb[0].template store<BM>(a.B, reinterpret_cast<float *>(ldsB));
      } else {
        sa[DEPTH - 1].template store<AM>(a.A, reinterpret_cast<float *>(ldsA));
        sb[DEPTH - 1].template store<BM>(a.B, reinterpret_cast<float *>(ldsB));
      }
    } else {
      if (set == 0 || DEPTH == 1) {
        sa[0].template store<AM>(a.A, reinterpret_cast<uint4 *>(ldsA));
        sb[0].template store<BM>(a.B, reinterpret_cast<uint4 *>(ldsB));
      } else {
        sa[DEPTH - 1].template store<AM>(a.A, reinterpret_cast<uint4 *>(ldsA));
        sb[DEPTH - 1].template store<BM>(a.B, reinterpret_cast<uint4 *>(ldsB));
      }
    }
    __syncthreads();
  };
  if (g0 < g1) issue(0, g0);
  if constexpr (DEPTH == 2) {
    if (g0 + 1 < g1) issue(1, g0 + 1);
    for (int g = g0; g < g1; g += 2) {
      to_lds(0);
      if (g + 2 < g1) issue(0, g + 2);
      multiply();
      if (g + 1 < g1) {
        to_lds(1);
        if (g + 3 < g1) issue(1, g + 3);
        multiply();
      }
    }
  } else {
    for (int g = g0; g < g1; ++g) {
      to_lds(0);
      if (g + 1 < g1) issue(0, g + 1);
      multiply();
    }
  }

  // D: a lane holds column j = lane & 31 and rows (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) of each 32x32 block
  const int I = a.A.R, J = a.B.R;
  const bool part = a.nsplit > 1;
  float *D = a.D + (part ? (long long)z * I * J : 0);
  const long long si = part ? J : a.d_si, sj = part ? 1 : a.d_sj;
#pragma unroll
  for (int x = 0; x < WI; ++x)
#pragma unroll
    for (int y = 0; y < WJ; ++y) {
      const int j = j0 + wj0 + 32 * y + lr;
      long long joff = j * sj;
      if (!part && a.d_fold_n) {
        const int b = j / a.d_fold_n;
        joff = b * a.d_sb + (j - b * a.d_fold_n) * sj;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = i0 + wi0 + 32 * x + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (i < I && j < J) {
          float v = acc[x][y][e];
          if (a.ep_scale) v = __builtin_fmaf(v, a.ep_scale[i], a.ep_shift[i]);
          if (a.ep_res) v += a.ep_res[i * si + joff];
          if (a.ep_relu) v = v > 0.f ? v : 0.f;
          D[i * si + joff] = v;
        }
      }
    }
}

// ---- slice-ordered sum of the split-K partials.  A workgroup owns 16 consecutive output elements; its 16 thread
// groups take the slices s = g, g+16, g+32, ... and the 16 sub-sums are added in group order through LDS: the
// summation order depends on nsplit only (bit-reproducible), and 8 MB of partials are read by a few hundred
// workgroups instead of a few dozen.  MODE 1 also turns d wcat [2Co, C] (+ d W_r) into d W [Co, 3+2C].
template <int MODE>
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float *__restrict__ part, int nsplit, int IJ,
                                                          float *__restrict__ out, const float *__restrict__ dwr,
                                                          int Co, int C) {
  __shared__ float s_sum[2][16][17];
  const int el = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const int n_el = MODE == 0 ? IJ : Co * C;
  for (int e0 = blockIdx.x * 16; e0 < n_el; e0 += gridDim.x * 16) {
    const int e = e0 + el;
    float top = 0.f, bot = 0.f;
    if (e < n_el) {
      if (MODE == 0) {
        for (int p = sg; p < nsplit; p += 16) top += part[(size_t)p * IJ + e];
      } else {  // e = (o, c) over [Co, C]: top = d wcat[o][c], bot = d wcat[Co + o][c]
        for (int p = sg; p < nsplit; p += 16) {
          top += part[(size_t)p * IJ + e];
          bot += part[(size_t)p * IJ + (size_t)Co * C + e];
        }
      }
    }
    s_sum[0][sg][el] = top;
    s_sum[1][sg][el] = bot;
    __syncthreads();
    if (sg == 0 && e < n_el) {
      top = s_sum[0][0][el];
      bot = s_sum[1][0][el];
#pragma unroll
      for (int g = 1; g < 16; ++g) {
        top += s_sum[0][g][el];
        bot += s_sum[1][g][el];
      }
      if (MODE == 0) {
        out[e] = top;
      } else {
        const int o = e / C, c = e - o * C;
        const int ld = 3 + 2 * C;
        out[(size_t)o * ld + 3 + c] = bot;            // d W_c
        out[(size_t)o * ld + 3 + C + c] = top - bot;  // d W_d
        if (c < 3) out[(size_t)o * ld + c] = dwr ? dwr[o * 3 + c] : 0.f;
      }
    }
    __syncthreads();
  }
}

// W [Co, 3+2C] = [W_r | W_c | W_d]  ->  wr [Co,3] and the per-point GEMM weight wcat [2Co, C] = [W_d ; W_c - W_d]
// (one small launch per forward pass; both gradients read wcat again)
__global__ __launch_bounds__(256) void pwmlp_weights_kernel(const float *__restrict__ W, int Co, int C,
                                                            float *__restrict__ wr, float *__restrict__ wcat) {
  const int ld = 3 + 2 * C;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < Co * ld; e += gridDim.x * 256) {
    const int o = e / ld, k = e - o * ld;
    const float v = W[e];
    if (k < 3) {
      if (wr) wr[o * 3 + k] = v;
    } else if (k < 3 + C) {
      wcat[(size_t)(Co + o) * C + (k - 3)] = v - W[e + C];
    } else {
      wcat[(size_t)o * C + (k - 3 - C)] = v;
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------
static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// operand whose r and k are plain strided axes
static GemmOperand plain(const float *p, long long sr, long long sk, int R, int K) {
  GemmOperand s{};
  s.p = p; s.sr = (int)sr; s.sk = (int)sk; s.R = R;
  s.rc = (sr == 1) ? 1 : 0;
  if (s.rc) s.vec = aligned16(p) && R % 4 == 0 && sk % 4 == 0;
  else s.vec = aligned16(p) && K % 4 == 0 && sr % 4 == 0;
  return s;
}

// channel-major tensor [nb, rows, n]: `point_on_r` puts the folded (cloud, point) axis on r and the rows on k, or
// the other way round
static GemmOperand channel_major(const float *p, int nb, int rows, int n, bool point_on_r) {
  GemmOperand s{};
  s.p = p; s.fold_n = n; s.sb = (long long)rows * n;
  if (point_on_r) {
    s.sr = 1; s.sk = n; s.R = nb * n; s.rc = 1; s.fold = nb > 1 ? 1 : 0;
  } else {
    s.sr = n; s.sk = 1; s.R = rows; s.rc = 0; s.fold = nb > 1 ? 2 : 0;
  }
  s.vec = aligned16(p) && n % 4 == 0;
  return s;
}

static int stage_mode(const GemmOperand &s) { return !s.vec ? STAGE_SCALAR : (s.rc ? STAGE_VEC_RC : STAGE_VEC_KC); }

template <int PREC, int AM, int BM>
static void launch_shape(const GemmArgs &a, int wi, int wj, int blocks, hipStream_t st) {
  if (wi == 2 && wj == 2) hipLaunchKernelGGL((mfma_gemm_kernel<PREC, 2, 2, AM, BM>), dim3(blocks), dim3(256), 0, st, a);
  else if (wi == 2) hipLaunchKernelGGL((mfma_gemm_kernel<PREC, 2, 1, AM, BM>), dim3(blocks), dim3(256), 0, st, a);
  else if (wj == 2) hipLaunchKernelGGL((mfma_gemm_kernel<PREC, 1, 2, AM, BM>), dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((mfma_gemm_kernel<PREC, 1, 1, AM, BM>), dim3(blocks), dim3(256), 0, st, a);
}

// the operand-mode pairs the entry points below produce (anything else, e.g. N % 4 != 0, takes the scalar pair)
template <int PREC>
static void launch_modes(GemmArgs &a, int wi, int wj, int blocks, hipStream_t st) {
  const int am = stage_mode(a.A), bm = stage_mode(a.B);
#define CL3D_GEMM_CASE(AM, BM) \
  if (am == AM && bm == BM) return launch_shape<PREC, AM, BM>(a, wi, wj, blocks, st);
  CL3D_GEMM_CASE(STAGE_VEC_RC, STAGE_VEC_KC)   // the three point GEMMs: features x wcat, wcat^T x d ght, d ght x features
  CL3D_GEMM_CASE(STAGE_VEC_KC, STAGE_VEC_RC)   // conv forward: W x x
  CL3D_GEMM_CASE(STAGE_VEC_RC, STAGE_VEC_RC)   // conv d x: W^T x d y
  CL3D_GEMM_CASE(STAGE_VEC_KC, STAGE_VEC_KC)   // conv d weight: d y x x
#undef CL3D_GEMM_CASE
  a.A.vec = a.B.vec = 0;
  launch_shape<PREC, STAGE_SCALAR, STAGE_SCALAR>(a, wi, wj, blocks, st);
}

constexpr int kCUs = 256;

// Workgroup tile for an I x J output (x nsplit K slices).  A workgroup's four waves sit on the four SIMDs of a CU,
// so a CU works through its workgroups' MFMA chains one after the other: time ~ rounds x (accumulators per wave),
// rounds = workgroups / CUs.  Fractional rounds below one count as one (the chain is serial per wave); ties go to
// the larger tile (fewer re-staged operand bytes).
static void pick_tile(int I, int J, int nsplit, int *wi_out, int *wj_out) {
  if (const char *force = getenv("CL3D_GEMM_TILE")) {  // tuning override "wi,wj" (scripts/bench_point_gemm.py --tiles)
    int wi = 0, wj = 0;
    if (sscanf(force, "%d,%d", &wi, &wj) == 2 && (wi == 1 || wi == 2) && (wj == 1 || wj == 2)) {
      *wi_out = wi;
      *wj_out = wj;
      return;
    }
  }
  const int cand[4][2] = {{2, 2}, {2, 1}, {1, 2}, {1, 1}};
  double best = 1e300;
  for (int c = 0; c < 4; ++c) {
    const int wi = cand[c][0], wj = cand[c][1];
    const long long blocks = (long long)ceil_div(I, 64 * wi) * ceil_div(J, 64 * wj) * (nsplit > 1 ? nsplit : 1);
    double rounds = (double)blocks / kCUs;
    if (rounds < 1.0) rounds = 1.0;
    const double cost = rounds * wi * wj;
    if (cost < best * 0.97) {
      best = cost;
      *wi_out = wi;
      *wj_out = wj;
    }
  }
}

static int kc_of(int precision) { return precision == PREC_BF16 ? 64 : 32; }

// K slices of a weight-gradient contraction: about one workgroup per CU over all output tiles, at least 8 chunks per
// slice so that a partial tile is written once per >= 256 points
static int wgrad_slices(long long K, int kc, int I, int J, int *chunks_per_split) {
  const long long total = (K + kc - 1) / kc;
  const int tiles = ceil_div(I, 128) * ceil_div(J, 64);  // the 128 x 64 tile, as run_gemm picks for small outputs
  long long want = kCUs / (tiles > 0 ? tiles : 1);
  if (want < 1) want = 1;
  long long cps = (total + want - 1) / want;
  long long min_cps = 8;
  if (const char *force = getenv("CL3D_GEMM_MIN_CPS")) min_cps = atoll(force) > 0 ? atoll(force) : 8;  // tuning override
  if (cps < min_cps) cps = min_cps;
  if (cps > total) cps = total > 0 ? total : 1;
  *chunks_per_split = (int)cps;
  return (int)((total + cps - 1) / cps);
}

static int run_gemm(GemmArgs &a, int precision, hipStream_t st, const char *who) {
  const int I = a.A.R, J = a.B.R;
  int wi = 2, wj = 2;
  pick_tile(I, J, a.nsplit, &wi, &wj);
  a.tiles_i = ceil_div(I, 64 * wi);
  a.tiles_j = ceil_div(J, 64 * wj);
  const long long blocks = (long long)a.tiles_i * a.tiles_j * (a.nsplit > 1 ? a.nsplit : 1);
  if (blocks > 0x7fffffffLL) return fail(CL3D_E_UNSUPPORTED, "%s: grid too large", who);
  if (blocks == 0) return CL3D_OK;
  if (precision == PREC_BF16) launch_modes<PREC_BF16>(a, wi, wj, (int)blocks, st);
  else launch_modes<PREC_F32>(a, wi, wj, (int)blocks, st);
  return check_launch(who);
}

static int round_up_grid(int n) {
  const int g = ceil_div(n, 256);
  return g < 1 ? 1 : (g > 2048 ? 2048 : g);
}

size_t gemm_wgrad_workspace(int nb, int K, int I, int J) {
  size_t worst = 0;
  for (int prec = 0; prec < 2; ++prec) {
    int cps = 0;
    const int ns = wgrad_slices((long long)nb * K, kc_of(prec), I, J, &cps);
    const size_t bytes = (size_t)ns * I * J * sizeof(float);
    worst = bytes > worst ? bytes : worst;
  }
  return worst;
}

// weight gradient D[I][J] = sum over all nb*K points: slices to scratch, then the ordered reduce
template <int MODE>
static int run_wgrad(GemmArgs &a, int precision, int nb, int K, void *ws, size_t ws_bytes, float *out, const float *dwr,
                     int Co, int C, hipStream_t st, const char *who) {
  const int I = a.A.R, J = a.B.R;
  const size_t need = gemm_wgrad_workspace(nb, K, I, J);
  if (!ws || ws_bytes < need) return fail(CL3D_E_WORKSPACE, "%s: workspace %zu < %zu", who, ws_bytes, need);
  a.K = nb * K;
  a.nsplit = wgrad_slices((long long)nb * K, kc_of(precision), I, J, &a.chunks_per_split);
  float *partial = static_cast<float *>(ws);
  if (a.nsplit == 1) a.nsplit = 0;  // one slice: the "partial" is the result, still through the reduce for MODE 1
  a.D = partial; a.d_si = J; a.d_sj = 1; a.d_fold_n = 0;
  int rc = run_gemm(a, precision, st, who);
  if (rc != CL3D_OK) return rc;
  const int n_el = MODE == 0 ? I * J : Co * C;
  int grid = ceil_div(n_el, 16);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL((gemm_reduce_kernel<MODE>), dim3(grid), dim3(256), 0, st, partial, a.nsplit > 1 ? a.nsplit : 1, I * J,
                     out, dwr, Co, C);
  return check_launch(who);
}

}  // namespace cl3d

using namespace cl3d;

#define GEMM_COMMON_CHECKS(who)                                                                       \
  CL3D_REQUIRE(B >= 0 && C >= 1 && N >= 1 && Co >= 1, who ": bad sizes");                            \
  CL3D_REQUIRE(precision == 0 || precision == 1, who ": precision must be 0 (f32) or 1 (bf16)");      \
  if ((long long)B * N * (long long)(2 * Co > C ? 2 * Co : C) > 0x7fffffffLL)                         \
    return fail(CL3D_E_UNSUPPORTED, who ": tensor too large (32-bit element offsets)");

extern "C" int cl3d_pwmlp_point_gemm_fwd(const float *features, const float *W, int B, int C, int N, int Co,
                                         int precision, float *ght, float *wr, float *wcat, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("pwmlp_point_gemm_fwd");
  CL3D_REQUIRE(W && wcat && (B == 0 || (features && ght)), "pwmlp_point_gemm_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(pwmlp_weights_kernel, dim3(round_up_grid(Co * (3 + 2 * C))), dim3(256), 0, st, W, Co, C, wr, wcat);
  const int rc = check_launch("cl3d_pwmlp_point_gemm_fwd(weights)");
  if (rc != CL3D_OK || B == 0) return rc;
  GemmArgs a{};  // D[i = (cloud, point)][j = o] = sum_c F[c][point] wcat[o][c]  ->  ght [B*N, 2Co]
  a.A = channel_major(features, B, C, N, true);
  a.B = plain(wcat, C, 1, 2 * Co, C);
  a.D = ght; a.d_si = 2 * Co; a.d_sj = 1;
  a.K = C;
  return run_gemm(a, precision, st, "cl3d_pwmlp_point_gemm_fwd");
}

extern "C" int cl3d_pwmlp_point_gemm_bwd_data(const float *dght, const float *wcat, int B, int C, int N, int Co,
                                              int precision, float *dfeatures, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("pwmlp_point_gemm_bwd_data");
  CL3D_REQUIRE(wcat && (B == 0 || (dght && dfeatures)), "pwmlp_point_gemm_bwd_data: null pointer");
  if (B == 0) return CL3D_OK;
  GemmArgs a{};  // D[i = c][j = (cloud, point)] = sum_o wcat[o][c] dght[point][o]  ->  d features [B, C, N]
  a.A = plain(wcat, 1, C, C, 2 * Co);
  a.B = plain(dght, 2 * Co, 1, B * N, 2 * Co);
  a.D = dfeatures; a.d_si = N; a.d_sj = 1; a.d_fold_n = B > 1 ? N : 0; a.d_sb = (long long)C * N;
  a.K = 2 * Co;
  return run_gemm(a, precision, (hipStream_t)stream, "cl3d_pwmlp_point_gemm_bwd_data");
}

extern "C" int cl3d_pwmlp_point_gemm_bwd_weight(const float *features, const float *dght, const float *dwr, int B,
                                                int C, int N, int Co, int precision, float *dW, void *ws,
                                                size_t ws_bytes, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("pwmlp_point_gemm_bwd_weight");
  CL3D_REQUIRE(B >= 1 && features && dght && dW, "pwmlp_point_gemm_bwd_weight: null pointer");
  GemmArgs a{};  // D[i = o][j = c] = sum_(cloud, point) dght[point][o] F[c][point]  ->  d wcat [2Co, C] per slice
  a.A = plain(dght, 1, 2 * Co, 2 * Co, B * N);
  a.B = channel_major(features, B, C, N, false);
  return run_wgrad<1>(a, precision, B, N, ws, ws_bytes, dW, dwr, Co, C, (hipStream_t)stream,
                      "cl3d_pwmlp_point_gemm_bwd_weight");
}

// ---- the 1x1 Conv1d layers around the operator (backbones/resnet.py:32-39,58-66), channel-major in and out ---------
extern "C" int cl3d_conv1x1_fwd(const float *x, const float *W, int B, int C, int N, int Co, int precision, float *y,
                                cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("conv1x1_fwd");
  CL3D_REQUIRE(W && (B == 0 || (x && y)), "conv1x1_fwd: null pointer");
  if (B == 0) return CL3D_OK;
  GemmArgs a{};  // D[i = o][j = (cloud, point)] = sum_c W[o][c] x[c][point]
  a.A = plain(W, C, 1, Co, C);
  a.B = channel_major(x, B, C, N, true);
  a.D = y; a.d_si = N; a.d_sj = 1; a.d_fold_n = B > 1 ? N : 0; a.d_sb = (long long)Co * N;
  a.K = C;
  return run_gemm(a, precision, (hipStream_t)stream, "cl3d_conv1x1_fwd");
}

// inference: y = act(scale[o] * (W x)[o] + shift[o] + residual) in the GEMM's epilogue -- the BatchNorm of eval mode
// folded into a per-channel affine map (scale = gamma / sqrt(var + eps), shift = beta - mean * scale), the shortcut
// add and the ReLU of backbones/resnet.py:58-66 without a pass of their own
extern "C" int cl3d_conv1x1_bn_act_fwd(const float *x, const float *W, const float *scale, const float *shift,
                                       const float *residual, int relu, int B, int C, int N, int Co, int precision,
                                       float *y, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("conv1x1_bn_act_fwd");
  CL3D_REQUIRE(W && (B == 0 || (x && y)) && (!scale == !shift), "conv1x1_bn_act_fwd: null pointer");
  if (B == 0) return CL3D_OK;
  GemmArgs a{};
  a.A = plain(W, C, 1, Co, C);
  a.B = channel_major(x, B, C, N, true);
  a.D = y; a.d_si = N; a.d_sj = 1; a.d_fold_n = B > 1 ? N : 0; a.d_sb = (long long)Co * N;
  a.K = C;
  a.ep_scale = scale; a.ep_shift = shift; a.ep_res = residual; a.ep_relu = relu;
  return run_gemm(a, precision, (hipStream_t)stream, "cl3d_conv1x1_bn_act_fwd");
}

extern "C" int cl3d_conv1x1_bwd_data(const float *dy, const float *W, int B, int C, int N, int Co, int precision,
                                     float *dx, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("conv1x1_bwd_data");
  CL3D_REQUIRE(W && (B == 0 || (dy && dx)), "conv1x1_bwd_data: null pointer");
  if (B == 0) return CL3D_OK;
  GemmArgs a{};  // D[i = c][j = (cloud, point)] = sum_o W[o][c] dy[o][point]
  a.A = plain(W, 1, C, C, Co);
  a.B = channel_major(dy, B, Co, N, true);
  a.D = dx; a.d_si = N; a.d_sj = 1; a.d_fold_n = B > 1 ? N : 0; a.d_sb = (long long)C * N;
  a.K = Co;
  return run_gemm(a, precision, (hipStream_t)stream, "cl3d_conv1x1_bwd_data");
}

extern "C" int cl3d_conv1x1_bwd_weight(const float *x, const float *dy, int B, int C, int N, int Co, int precision,
                                       float *dW, void *ws, size_t ws_bytes, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("conv1x1_bwd_weight");
  CL3D_REQUIRE(B >= 1 && x && dy && dW, "conv1x1_bwd_weight: null pointer");
  GemmArgs a{};  // D[i = o][j = c] = sum_(cloud, point) dy[o][point] x[c][point]
  a.A = channel_major(dy, B, Co, N, false);
  a.B = channel_major(x, B, C, N, false);
  return run_wgrad<0>(a, precision, B, N, ws, ws_bytes, dW, nullptr, Co, C, (hipStream_t)stream, "cl3d_conv1x1_bwd_weight");
}

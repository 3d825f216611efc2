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
//   * a 256-thread workgroup = 2 x 2 waves, each wave a WI x WJ grid of 32x32 accumulators (64 x 64 per wave for
//     the large shapes, 128 x 128 per workgroup); the next K chunk is prefetched into registers while the current
//     one is multiplied out of LDS;
//   * weight gradients contract over ALL points (K = B*N): the (batch, point) axis is cut into contiguous slices,
//     one workgroup each, partial products go to scratch and are summed in slice order by a second small kernel
//     (no atomics: bit-reproducible), which also folds the PointWiseMLP weight plumbing
//     (d W_c = bot, d W_d = top - bot, d W_r) so no separate merge pass exists on this path.
// The PointWiseMLP weight [Co, 3+2C] = [W_r | W_c | W_d] is turned into wcat = [W_d ; W_c - W_d] and W_r by one small
// launch ahead of the forward GEMM (pwmlp_weights_kernel); the weight-gradient reduce writes d W directly.
#include "cl3d_common.h"

namespace cl3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { PREC_F32 = 0, PREC_BF16 = 1 };
enum { STAGE_VEC_RC = 0, STAGE_VEC_KC = 1, STAGE_SCALAR = 2 };  // how an operand tile travels HBM -> LDS

struct GemmOperand {
  const float *p;
  int sr, sk;            // element strides along the tile index r and the contraction index k
  long long sb;          // ... and the batch
  int R;                 // extent along r
  int rc;                // 1: r is the contiguous axis (sr == 1), 0: k is (sk == 1)
  int vec;               // 16-byte loads along the contiguous axis are legal (alignment and extents)
};

struct GemmArgs {
  GemmOperand A, B;  // D[i][j] = sum_k A(i,k) B(j,k)
  float *D;
  long long d_si, d_sj, d_sb;
  int K;               // contraction extent per batch
  int nb;              // batches
  int split;           // 0: grid = tiles x batches; 1: (batch, k) folded, cut into nsplit slices, D += slice*I*J
  int nsplit, chunks_per_split;
  int tiles_i, tiles_j;
};

// element (r, k) of an operand relative to the tile origin `base` (wave-uniform: scalar base + 32-bit lane offset)
__device__ __forceinline__ float gemm_fetch(const GemmOperand &s, const float *base, int r, int k) {
  return base[(unsigned)(r * s.sr + k * s.sk)];
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
  __device__ __forceinline__ void load(const GemmOperand &s, long long boff, int r0, int k0, int K) {
    const int t = threadIdx.x;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *base = s.p + boff + (long long)r0 * s.sr + (long long)k0 * s.sk;
    const int Rl = s.R - r0, Kl = K - k0;  // what is left of the operand from the tile origin on
    if (MODE != STAGE_SCALAR) {
      if (MODE == STAGE_VEC_RC) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int idx = q * 256 + t, r = 4 * (idx % (TR / 4)), k = idx / (TR / 4);
          v[q] = (r < Rl && k < Kl) ? ld4(base + (unsigned)(k * s.sk + r)) : zero;
        }
      } else {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int idx = q * 256 + t, k = 4 * (idx % (KC / 4)), r = idx / (KC / 4);
          v[q] = (r < Rl && k < Kl) ? ld4(base + (unsigned)(r * s.sr + k)) : zero;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = (q * 4 + e) * 256 + t;
          const int r = s.rc ? idx % TR : idx / KC, k = s.rc ? idx / TR : idx % KC;
          comp(v[q], e) = (r < Rl && k < Kl) ? gemm_fetch(s, base, r, k) : 0.f;
        }
    }
  }

  template <int MODE>
  __device__ __forceinline__ void store(const GemmOperand &s, float *T) const {
    const int t = threadIdx.x;
    if (MODE != STAGE_SCALAR) {
      if (MODE == STAGE_VEC_RC) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int idx = q * 256 + t;
          *reinterpret_cast<float4 *>(T + (idx / (TR / 4)) * (TR + 4) + 4 * (idx % (TR / 4))) = v[q];
        }
      } else {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int idx = q * 256 + t, k = 4 * (idx % (KC / 4)), r = idx / (KC / 4);
          float4 x = v[q];
#pragma unroll
          for (int e = 0; e < 4; ++e) T[(k + e) * (TR + 1) + r] = comp(x, e);
        }
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
  __device__ __forceinline__ void load(const GemmOperand &s, long long boff, int r0, int k0, int K) {
    const int t = threadIdx.x;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *base = s.p + boff + (long long)r0 * s.sr + (long long)k0 * s.sk;
    const int Rl = s.R - r0, Kl = K - k0;
    if (MODE == STAGE_VEC_RC) {  // one (k group, 4 rows) item per thread: 8 loads, each coalesced over the wave
      const int g = t / (TR / 4), r = 4 * (t % (TR / 4));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 8 * g + e;
        v[e] = (g < G && r < Rl && k < Kl) ? ld4(base + (unsigned)(k * s.sk + r)) : zero;
      }
    } else if (MODE == STAGE_VEC_KC) {  // (row, k group) items: 32 contiguous bytes each, 8 lanes cover a 256-byte run of a row
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int idx = q * 256 + t, g = idx % G, r = idx / G, k = 8 * g;
        const bool ok = idx < TR * G && r < Rl;
        const float *src = base + (unsigned)(r * s.sr + k);
        v[2 * q] = (ok && k < Kl) ? ld4(src) : zero;
        v[2 * q + 1] = (ok && k + 4 < Kl) ? ld4(src + 4) : zero;
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int idx = q * 256 + t;
        const int g = s.rc ? idx / TR : idx % G, r = s.rc ? idx % TR : idx / G;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 8 * g + e;
          comp(v[2 * q + e / 4], e % 4) = (idx < TR * G && r < Rl && k < Kl) ? gemm_fetch(s, base, r, k) : 0.f;
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

  // tile / batch / slice of this workgroup
  int bid = blockIdx.x;
  const int tj = bid % a.tiles_j;
  bid /= a.tiles_j;
  const int ti = bid % a.tiles_i;
  const int z = bid / a.tiles_i;  // batch (split == 0) or K slice (split == 1)
  const int i0 = ti * TI, j0 = tj * TJ;
  const int cpb = (a.K + KC - 1) / KC;  // chunks per batch
  int g0, g1;
  if (a.split) {
    g0 = z * a.chunks_per_split;
    g1 = g0 + a.chunks_per_split;
    const int total = cpb * a.nb;
    if (g1 > total) g1 = total;
  } else {
    g0 = z * cpb;
    g1 = g0 + cpb;
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

  typename SA::type sa;
  typename SB::type sb;
  if (g0 < g1) {
    const int b = g0 / cpb, k0 = (g0 - b * cpb) * KC;
    sa.template load<AM>(a.A, (long long)b * a.A.sb, i0, k0, a.K);
    sb.template load<BM>(a.B, (long long)b * a.B.sb, j0, k0, a.K);
  }
  for (int g = g0; g < g1; ++g) {
    __syncthreads();  // the previous chunk has been multiplied out of LDS
    if constexpr (PREC == PREC_F32) {
      sa.template store<AM>(a.A, reinterpret_cast<float *>(ldsA));
      sb.template store<BM>(a.B, reinterpret_cast<float *>(ldsB));
    } else {
      sa.template store<AM>(a.A, reinterpret_cast<uint4 *>(ldsA));
      sb.template store<BM>(a.B, reinterpret_cast<uint4 *>(ldsB));
    }
    __syncthreads();
    if (g + 1 < g1) {  // next chunk's loads stay in flight while this one is multiplied
      const int b = (g + 1) / cpb, k0 = (g + 1 - b * cpb) * KC;
      sa.template load<AM>(a.A, (long long)b * a.A.sb, i0, k0, a.K);
      sb.template load<BM>(a.B, (long long)b * a.B.sb, j0, k0, a.K);
    }
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
  }

  // D: a lane holds column j = lane & 31 and rows (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) of each 32x32 block
  const int I = a.A.R, J = a.B.R;
  float *D = a.D + (a.split ? (long long)z * I * J : (long long)z * a.d_sb);
  const long long si = a.split ? J : a.d_si, sj = a.split ? 1 : a.d_sj;
#pragma unroll
  for (int x = 0; x < WI; ++x)
#pragma unroll
    for (int y = 0; y < WJ; ++y) {
      const int j = j0 + wj0 + 32 * y + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = i0 + wi0 + 32 * x + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (i < I && j < J) D[i * si + j * sj] = acc[x][y][e];
      }
    }
}

// ---- slice-ordered sum of the split-K partials; MODE 1 also turns d wcat [2Co, C] (+ d W_r) into d W [Co, 3+2C] --
template <int MODE>
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float *__restrict__ part, int nsplit, int IJ,
                                                          float *__restrict__ out, const float *__restrict__ dwr,
                                                          int Co, int C) {
  for (int e = blockIdx.x * 256 + threadIdx.x; e < IJ; e += gridDim.x * 256) {
    if (MODE == 0) {
      float s = 0.f;
      for (int p = 0; p < nsplit; ++p) s += part[(size_t)p * IJ + e];
      out[e] = s;
    } else {  // e = (o, c) over [Co, C]: top = d wcat[o][c], bot = d wcat[Co + o][c]
      if (e >= Co * C) continue;
      const int o = e / C, c = e - o * C;
      float top = 0.f, bot = 0.f;
      for (int p = 0; p < nsplit; ++p) {
        top += part[(size_t)p * IJ + (size_t)o * C + c];
        bot += part[(size_t)p * IJ + (size_t)(Co + o) * C + c];
      }
      const int ld = 3 + 2 * C;
      out[(size_t)o * ld + 3 + c] = bot;            // d W_c
      out[(size_t)o * ld + 3 + C + c] = top - bot;  // d W_d
      if (c < 3) out[(size_t)o * ld + c] = dwr ? dwr[o * 3 + c] : 0.f;
    }
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

static GemmOperand plain(const float *p, long long sr, long long sk, long long sb, int R, int K) {
  GemmOperand s{};
  s.p = p; s.sr = (int)sr; s.sk = (int)sk; s.sb = sb; s.R = R;
  s.rc = (sr == 1) ? 1 : 0;
  if (s.rc) s.vec = aligned16(p) && R % 4 == 0 && sk % 4 == 0 && sb % 4 == 0;
  else s.vec = aligned16(p) && K % 4 == 0 && sr % 4 == 0 && sb % 4 == 0;
  return s;
}

static int stage_mode(const GemmOperand &s) { return !s.vec ? STAGE_SCALAR : (s.rc ? STAGE_VEC_RC : STAGE_VEC_KC); }

template <int PREC, int AM, int BM>
static void launch_shape(const GemmArgs &a, int wi, int wj, int blocks, hipStream_t st) {
  if (wi == 2 && wj == 2) hipLaunchKernelGGL((mfma_gemm_kernel<PREC, 2, 2, AM, BM>), dim3(blocks), dim3(256), 0, st, a);
  else if (wi == 2) hipLaunchKernelGGL((mfma_gemm_kernel<PREC, 2, 1, AM, BM>), dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((mfma_gemm_kernel<PREC, 1, 2, AM, BM>), dim3(blocks), dim3(256), 0, st, a);
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

// number of K slices of a weight-gradient contraction over nb batches of K points each
static int wgrad_slices(int nb, int K, int kc, int tiles, int *chunks_per_split) {
  const long long total = (long long)nb * ceil_div(K, kc);
  long long want = 1024 / (tiles > 0 ? tiles : 1);  // ~4 workgroups per CU over all output tiles
  if (want < 1) want = 1;
  long long cps = (total + want - 1) / want;
  if (cps < 4) cps = 4;  // at least 4 chunks per slice: the partial tile is written once per slice
  if (cps > total) cps = total;
  *chunks_per_split = (int)cps;
  return (int)((total + cps - 1) / cps);
}

static int run_gemm(GemmArgs &a, int precision, bool split, hipStream_t st, const char *who) {
  const int I = a.A.R, J = a.B.R;
  int wi = I > 64 ? 2 : 1, wj = J > 64 ? 2 : 1;
  if (wi == 1 && wj == 1) wi = 2;  // no 64 x 64 workgroup tile: small outputs ride in the 128 x 64 one
  a.tiles_i = ceil_div(I, 64 * wi);
  a.tiles_j = ceil_div(J, 64 * wj);
  a.split = split ? 1 : 0;
  const long long blocks = (long long)a.tiles_i * a.tiles_j * (split ? a.nsplit : a.nb);
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

static int kc_of(int precision) { return precision == PREC_BF16 ? 64 : 32; }

static int out_tiles(int I, int J) {
  int wi = I > 64 ? 2 : 1, wj = J > 64 ? 2 : 1;
  if (wi == 1 && wj == 1) wi = 2;
  return ceil_div(I, 64 * wi) * ceil_div(J, 64 * wj);
}

size_t gemm_wgrad_workspace(int nb, int K, int I, int J) {
  size_t worst = 0;
  for (int prec = 0; prec < 2; ++prec) {
    int cps = 0;
    const int tiles = out_tiles(I, J);
    const int ns = wgrad_slices(nb, K, kc_of(prec), tiles, &cps);
    const size_t bytes = (size_t)ns * I * J * sizeof(float);
    worst = bytes > worst ? bytes : worst;
  }
  return worst;
}

}  // namespace cl3d

using namespace cl3d;

#define GEMM_COMMON_CHECKS(who)                                                                       \
  CL3D_REQUIRE(B >= 0 && C >= 1 && N >= 1 && Co >= 1, who ": bad sizes");                            \
  CL3D_REQUIRE(precision == 0 || precision == 1, who ": precision must be 0 (f32) or 1 (bf16)");      \
  if ((long long)B * N * (long long)(2 * Co > C ? 2 * Co : C) > 0x7fffffffffLL)                       \
    return fail(CL3D_E_UNSUPPORTED, who ": tensor too large");

extern "C" int cl3d_pwmlp_point_gemm_fwd(const float *features, const float *W, int B, int C, int N, int Co,
                                         int precision, float *ght, float *wr, float *wcat, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("pwmlp_point_gemm_fwd");
  CL3D_REQUIRE(W && wcat && (B == 0 || (features && ght)), "pwmlp_point_gemm_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(pwmlp_weights_kernel, dim3(round_up_grid(Co * (3 + 2 * C))), dim3(256), 0, st, W, Co, C, wr, wcat);
  const int rc = check_launch("cl3d_pwmlp_point_gemm_fwd(weights)");
  if (rc != CL3D_OK || B == 0) return rc;
  GemmArgs a{};  // D[i = point][j = o] = sum_c F[c][point] wcat[o][c]  ->  ght [B, N, 2Co]
  a.A = plain(features, 1, N, (long long)C * N, N, C);
  a.B = plain(wcat, C, 1, 0, 2 * Co, C);
  a.D = ght; a.d_si = 2 * Co; a.d_sj = 1; a.d_sb = (long long)N * 2 * Co;
  a.K = C; a.nb = B;
  return run_gemm(a, precision, false, st, "cl3d_pwmlp_point_gemm_fwd");
}

extern "C" int cl3d_pwmlp_point_gemm_bwd_data(const float *dght, const float *wcat, int B, int C, int N, int Co,
                                              int precision, float *dfeatures, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("pwmlp_point_gemm_bwd_data");
  CL3D_REQUIRE(wcat && (B == 0 || (dght && dfeatures)), "pwmlp_point_gemm_bwd_data: null pointer");
  if (B == 0) return CL3D_OK;
  GemmArgs a{};  // D[i = c][j = point] = sum_o wcat[o][c] dght[point][o]  ->  d features [B, C, N]
  a.A = plain(wcat, 1, C, 0, C, 2 * Co);
  a.B = plain(dght, 2 * Co, 1, (long long)N * 2 * Co, N, 2 * Co);
  a.D = dfeatures; a.d_si = N; a.d_sj = 1; a.d_sb = (long long)C * N;
  a.K = 2 * Co; a.nb = B;
  return run_gemm(a, precision, false, (hipStream_t)stream, "cl3d_pwmlp_point_gemm_bwd_data");
}

extern "C" int cl3d_pwmlp_point_gemm_bwd_weight(const float *features, const float *dght, const float *dwr, int B,
                                                int C, int N, int Co, int precision, float *dW, void *ws,
                                                size_t ws_bytes, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("pwmlp_point_gemm_bwd_weight");
  CL3D_REQUIRE(B >= 1 && features && dght && dW, "pwmlp_point_gemm_bwd_weight: null pointer");
  const size_t need = gemm_wgrad_workspace(B, N, 2 * Co, C);
  if (!ws || ws_bytes < need) return fail(CL3D_E_WORKSPACE, "pwmlp_point_gemm_bwd_weight: workspace %zu < %zu", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a{};  // D[i = o][j = c] = sum_(b, point) dght[point][o] F[c][point]  ->  d wcat [2Co, C] per slice
  a.A = plain(dght, 1, 2 * Co, (long long)N * 2 * Co, 2 * Co, N);
  a.B = plain(features, N, 1, (long long)C * N, C, N);
  a.D = static_cast<float *>(ws);
  a.K = N; a.nb = B;
  const int tiles = out_tiles(2 * Co, C);
  a.nsplit = wgrad_slices(B, N, kc_of(precision), tiles, &a.chunks_per_split);
  int rc = run_gemm(a, precision, true, st, "cl3d_pwmlp_point_gemm_bwd_weight");
  if (rc != CL3D_OK) return rc;
  hipLaunchKernelGGL((gemm_reduce_kernel<1>), dim3(round_up_grid(Co * C)), dim3(256), 0, st, static_cast<const float *>(ws),
                     a.nsplit, 2 * Co * C, dW, dwr, Co, C);
  return check_launch("cl3d_pwmlp_point_gemm_bwd_weight(reduce)");
}

// ---- the 1x1 Conv1d layers around the operator (backbones/resnet.py:32-39,58-66), channel-major in and out ---------
extern "C" int cl3d_conv1x1_fwd(const float *x, const float *W, int B, int C, int N, int Co, int precision, float *y,
                                cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("conv1x1_fwd");
  CL3D_REQUIRE(W && (B == 0 || (x && y)), "conv1x1_fwd: null pointer");
  if (B == 0) return CL3D_OK;
  GemmArgs a{};  // D[i = o][j = point] = sum_c W[o][c] x[c][point]
  a.A = plain(W, C, 1, 0, Co, C);
  a.B = plain(x, 1, N, (long long)C * N, N, C);
  a.D = y; a.d_si = N; a.d_sj = 1; a.d_sb = (long long)Co * N;
  a.K = C; a.nb = B;
  return run_gemm(a, precision, false, (hipStream_t)stream, "cl3d_conv1x1_fwd");
}

extern "C" int cl3d_conv1x1_bwd_data(const float *dy, const float *W, int B, int C, int N, int Co, int precision,
                                     float *dx, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("conv1x1_bwd_data");
  CL3D_REQUIRE(W && (B == 0 || (dy && dx)), "conv1x1_bwd_data: null pointer");
  if (B == 0) return CL3D_OK;
  GemmArgs a{};  // D[i = c][j = point] = sum_o W[o][c] dy[o][point]
  a.A = plain(W, 1, C, 0, C, Co);
  a.B = plain(dy, 1, N, (long long)Co * N, N, Co);
  a.D = dx; a.d_si = N; a.d_sj = 1; a.d_sb = (long long)C * N;
  a.K = Co; a.nb = B;
  return run_gemm(a, precision, false, (hipStream_t)stream, "cl3d_conv1x1_bwd_data");
}

extern "C" int cl3d_conv1x1_bwd_weight(const float *x, const float *dy, int B, int C, int N, int Co, int precision,
                                       float *dW, void *ws, size_t ws_bytes, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("conv1x1_bwd_weight");
  CL3D_REQUIRE(B >= 1 && x && dy && dW, "conv1x1_bwd_weight: null pointer");
  const size_t need = gemm_wgrad_workspace(B, N, Co, C);
  if (!ws || ws_bytes < need) return fail(CL3D_E_WORKSPACE, "conv1x1_bwd_weight: workspace %zu < %zu", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a{};  // D[i = o][j = c] = sum_(b, point) dy[o][point] x[c][point]
  a.A = plain(dy, N, 1, (long long)Co * N, Co, N);
  a.B = plain(x, N, 1, (long long)C * N, C, N);
  a.D = static_cast<float *>(ws);
  a.K = N; a.nb = B;
  const int tiles = out_tiles(Co, C);
  a.nsplit = wgrad_slices(B, N, kc_of(precision), tiles, &a.chunks_per_split);
  int rc = run_gemm(a, precision, true, st, "cl3d_conv1x1_bwd_weight");
  if (rc != CL3D_OK) return rc;
  hipLaunchKernelGGL((gemm_reduce_kernel<0>), dim3(round_up_grid(Co * C)), dim3(256), 0, st, static_cast<const float *>(ws),
                     a.nsplit, Co * C, dW, nullptr, Co, C);
  return check_launch("cl3d_conv1x1_bwd_weight(reduce)");
}

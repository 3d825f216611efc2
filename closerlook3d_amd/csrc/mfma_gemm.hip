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
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <vector>

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
  int fold_shift;        // log2(fold_n) when it is a power of two (every level of a grid-subsampled pyramid that halves
                         // twice per stage), else -1: index -> (cloud, point) by shift instead of a division per access
  long long sb;          // cloud stride of a folded axis
  // prologue (nullable): element (r, k) enters the product as max(scale[c] * x + shift[c], 0) with c = k (pro_axis 0)
  // or c = r (pro_axis 1) -- the BatchNorm + ReLU that precedes this contraction in a bottleneck, applied while the
  // tile is staged instead of in a pass of its own (backbones/resnet.py:32-34,47-56: conv1's BatchNorm + ReLU in
  // front of the operator, the operator's in front of conv2)
  const float *pro_scale, *pro_shift;
  int pro_axis;
};

// where a result element (i, j) goes, and what happens to it on the way (inference epilogue)
struct OutMap {
  float *D;
  long long si, sj;
  int fold_n;          // != 0: j = cloud * fold_n + point, cloud stride sb (channel-major output)
  int fold_shift;      // log2(fold_n) or -1, as for the operands
  long long sb;
  // D = act(scale[i] * acc + shift[i] + res): BatchNorm folded to a per-row affine map, residual, ReLU
  const float *ep_scale, *ep_shift, *ep_res;
  int ep_relu;
};

struct GemmArgs {
  GemmOperand A, B;  // D[i][j] = sum_k A(i,k) B(j,k)
  OutMap out;
  float *partial;      // nsplit > 1: slice s writes its I x J tile (row-major) to partial + s*I*J instead
  int K;               // contraction extent (clouds folded in for weight gradients)
  int nsplit, chunks_per_split;
  int tiles_i, tiles_j;
  // nsplit > 1 and tickets != nullptr: the slices of a tile are summed INSIDE this launch -- every workgroup draws a
  // ticket for its tile after its partial is written; the one that draws the last adds the partials in slice order
  // 0, 1, 2, ... (whatever the order in which they arrived: bit-reproducible), sends the sums through `out` and puts
  // the ticket counter back to zero for the next launch.  No second launch, no 5 us hand-over between two kernels.
  unsigned *tickets;   // one counter per output tile, zero on entry and on exit
  int vec_out;         // the output map takes four consecutive j at once (16-byte stores; host-checked alignment)
};

__device__ __forceinline__ long long out_col(const OutMap &o, int j) {
  if (o.fold_n) {
    const int b = o.fold_shift >= 0 ? j >> o.fold_shift : j / o.fold_n;
    return b * o.sb + (long long)(j - b * o.fold_n) * o.sj;
  }
  return (long long)j * o.sj;
}
__device__ __forceinline__ void out_store(const OutMap &o, int i, long long joff, float v) {
  const long long at = i * o.si + joff;
  if (o.ep_scale) v = __builtin_fmaf(v, o.ep_scale[i], o.ep_shift[i]);
  if (o.ep_res) v += o.ep_res[at];
  if (o.ep_relu) v = v > 0.f ? v : 0.f;
  o.D[at] = v;
}

// element offset of (r, k) of an operand; the pointer base is wave-uniform, the offset a 32-bit lane value
__device__ __forceinline__ unsigned gemm_off(const GemmOperand &s, int r, int k) {
  if (s.fold == 1) {
    const int b = s.fold_shift >= 0 ? r >> s.fold_shift : r / s.fold_n;
    return (unsigned)(b * (int)s.sb + (r - b * s.fold_n) * s.sr + k * s.sk);
  }
  if (s.fold == 2) {
    const int b = s.fold_shift >= 0 ? k >> s.fold_shift : k / s.fold_n;
    return (unsigned)(b * (int)s.sb + r * s.sr + (k - b * s.fold_n) * s.sk);
  }
  return (unsigned)(r * s.sr + k * s.sk);
}

// Offsets of one load() call.  A tile starts at (r0, k0) with r0 % TR == 0 and k0 % KC == 0, so when the points per
// cloud are a multiple of the tile extent along the folded axis the whole tile lies in ONE cloud: the cloud index is
// one division per call on wave-uniform values instead of one per loaded vector (round 2: ~30 VALU instructions of
// address arithmetic per 16-byte load -- the f32 kernels issued 5-7 times the vendor kernels' VALU instructions).
struct TileOff {
  unsigned base;
  int rb, kb;
  bool fast;
};
template <int TR, int KC>
__device__ __forceinline__ TileOff tile_off(const GemmOperand &s, int r0, int k0) {
  TileOff t{0u, 0, 0, true};
  if (s.fold == 1) {
    t.fast = s.fold_n % TR == 0;
    if (t.fast) {
      const int b = r0 / s.fold_n;
      t.base = (unsigned)(b * (int)s.sb);
      t.rb = b * s.fold_n;
    }
  } else if (s.fold == 2) {
    t.fast = s.fold_n % KC == 0;
    if (t.fast) {
      const int b = k0 / s.fold_n;
      t.base = (unsigned)(b * (int)s.sb);
      t.kb = b * s.fold_n;
    }
  }
  return t;
}
__device__ __forceinline__ unsigned gemm_off(const GemmOperand &s, const TileOff &t, int r, int k) {
  if (t.fast) return t.base + (unsigned)((r - t.rb) * s.sr + (k - t.kb) * s.sk);
  return gemm_off(s, r, k);
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float &comp(float4 &v, int e) { return reinterpret_cast<float *>(&v)[e]; }

__device__ __forceinline__ float pro1(const GemmOperand &s, float x, int r, int k) {
  const int c = s.pro_axis ? r : k;
  const float z = __builtin_fmaf(x, s.pro_scale[c], s.pro_shift[c]);
  return z > 0.f ? z : 0.f;
}
// four elements starting at (r, k), running along r (ALONG_R) or along k; the operand's prologue, if any, applied
template <bool ALONG_R>
__device__ __forceinline__ float4 pro4(const GemmOperand &s, float4 v, int r, int k) {
  if (s.pro_scale == nullptr) return v;  // wave-uniform
  if ((s.pro_axis != 0) == ALONG_R) {    // the channel index runs along the vector: four channels
    const int c0 = ALONG_R ? r : k;
    const float4 sc = ld4(s.pro_scale + c0), sh = ld4(s.pro_shift + c0);  // c0 % 4 == 0 and 16-byte aligned arrays (host check)
    v.x = __builtin_fmaf(v.x, sc.x, sh.x); v.y = __builtin_fmaf(v.y, sc.y, sh.y);
    v.z = __builtin_fmaf(v.z, sc.z, sh.z); v.w = __builtin_fmaf(v.w, sc.w, sh.w);
  } else {                               // one channel for the whole vector
    const int c = ALONG_R ? k : r;
    const float sc = s.pro_scale[c], sh = s.pro_shift[c];
    v.x = __builtin_fmaf(v.x, sc, sh); v.y = __builtin_fmaf(v.y, sc, sh);
    v.z = __builtin_fmaf(v.z, sc, sh); v.w = __builtin_fmaf(v.w, sc, sh);
  }
  v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
  v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
  return v;
}

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
    const TileOff to = tile_off<TR, KC>(s, r0, k0);
    if (MODE == STAGE_VEC_RC) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int idx = q * 256 + t, r = r0 + 4 * (idx % (TR / 4)), k = k0 + idx / (TR / 4);
        v[q] = (r < s.R && k < K) ? pro4<true>(s, ld4(s.p + gemm_off(s, to, r, k)), r, k) : zero;
      }
    } else if (MODE == STAGE_VEC_KC) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int idx = q * 256 + t, k = k0 + 4 * (idx % (KC / 4)), r = r0 + idx / (KC / 4);
        v[q] = (r < s.R && k < K) ? pro4<false>(s, ld4(s.p + gemm_off(s, to, r, k)), r, k) : zero;
      }
    } else {
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = (q * 4 + e) * 256 + t;
          const int r = r0 + (s.rc ? idx % TR : idx / KC), k = k0 + (s.rc ? idx / TR : idx % KC);
          float x = (r < s.R && k < K) ? s.p[gemm_off(s, r, k)] : 0.f;
          if (s.pro_scale != nullptr && r < s.R && k < K) x = pro1(s, x, r, k);
          comp(v[q], e) = x;
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
// Row r of a k group sits at position swz(r) = r ^ ((r >> 4) & 3): the r-contiguous staging path has lane m write rows
// 4m .. 4m+3 one after the other, i.e. 16 lanes write 16-byte packs 64 bytes apart -- four lanes per bank group
// (SQ_LDS_BANK_CONFLICT 1.18e6 per launch in round 2's counters); with the swizzle lane m's i-th pack lands in
// bank group 4 (m % 4) + (i ^ (m >> 2)), all sixteen distinct.  The permutation stays inside blocks of four rows, so the
// 32 consecutive rows a fragment read covers still tile the banks exactly.
__device__ __forceinline__ int bf16_swz(int r) { return r ^ ((r >> 4) & 3); }

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
    const TileOff to = tile_off<TR, KC>(s, r0, k0);
    if (MODE == STAGE_VEC_RC) {  // one (k group, 4 rows) item per thread: 8 loads, each coalesced over the wave
      const int g = t / (TR / 4), r = r0 + 4 * (t % (TR / 4));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + 8 * g + e;
        v[e] = (g < G && r < s.R && k < K) ? pro4<true>(s, ld4(s.p + gemm_off(s, to, r, k)), r, k) : zero;
      }
    } else if (MODE == STAGE_VEC_KC) {  // (row, k group) items: 32 contiguous bytes, 8 lanes cover 256 bytes of a row
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int idx = q * 256 + t, g = idx % G, r = r0 + idx / G, k = k0 + 8 * g;
        const bool ok = idx < TR * G && r < s.R;
        v[2 * q] = (ok && k < K) ? pro4<false>(s, ld4(s.p + gemm_off(s, to, r, k)), r, k) : zero;
        v[2 * q + 1] = (ok && k + 4 < K) ? pro4<false>(s, ld4(s.p + gemm_off(s, to, r, k + 4)), r, k + 4) : zero;
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int idx = q * 256 + t;
        const int g = s.rc ? idx / TR : idx % G, r = r0 + (s.rc ? idx % TR : idx / G);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = k0 + 8 * g + e;
          const bool in = idx < TR * G && r < s.R && k < K;
          float x = in ? s.p[gemm_off(s, r, k)] : 0.f;
          if (s.pro_scale != nullptr && in) x = pro1(s, x, r, k);
          comp(v[2 * q + e / 4], e % 4) = x;
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
          T[g * (TR + 1) + bf16_swz(r + i)] = pack(x);
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
        if (idx < TR * G) T[g * (TR + 1) + bf16_swz(r)] = pack(x);
      }
    }
  }
};

// K chunk: f32 32 channels (16 MFMA k-steps per barrier round), bf16 64 (four 16-deep steps).  (48-deep chunks on the
// 64 x 64 tile -- fewer rounds, more bytes in flight per workgroup -- were measured on the config-2 backbone: 2 %
// slower; the larger LDS and register footprint costs a resident workgroup per CU.)
__host__ __device__ constexpr int gemm_kc(int prec, int wi, int wj) { return prec == PREC_BF16 ? 64 : ((void)wi, (void)wj, 32); }

template <int PREC, int TR, int KC>
struct StagePick {
  using type = StageF32<TR, KC>;
  static constexpr int kLdsBytes = StageF32<TR, KC>::kLdsFloats * 4;
};
template <int TR, int KC>
struct StagePick<PREC_BF16, TR, KC> {
  using type = StageBF16<TR>;
  static constexpr int kLdsBytes = StageBF16<TR>::kLdsPacks * 16;
};

// ---- K slices summed inside the launch.  Called by every workgroup of a sliced product after its
// partial tile has been STORED AT DEVICE SCOPE (write_tiles kind 2).  No fence anywhere: every wave waits for its own stores
// to be acknowledged (an explicit s_waitcnt vmcnt(0) in last_arrival), a device-scope store is acknowledged when it is
// visible to the other XCDs, the ticket is a device-scope atomic issued after the barrier, and the last arrival reads
// the other slices with device-scope loads (which do not hit whatever its own XCD's L2 still holds of the scratch).
__device__ __forceinline__ float ld_device_scope(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int TI, int TJ>
__device__ __forceinline__ void sum_slices_by_last_arrival(const GemmArgs &a, int i0, int j0, int tile) {
  // (cl3d_common.h: every wave waits for its slice stores to be acknowledged before the barrier in front of the ticket --
  //  round 6 found the barrier alone compiled to s_waitcnt vmcnt(63): no wait)
  __shared__ int s_last;
  if (!last_arrival(a.tickets + tile, (unsigned)a.nsplit, &s_last)) return;
  const int I = a.A.R, J = a.B.R;
  const long long IJ = (long long)I * J;
  const OutMap &o = a.out;
  if (a.vec_out) {  // J % 4 == 0: four consecutive j never straddle the matrix edge, a row or a cloud
    // a thread's NE four-element pieces of the tile are requested together for every slice (4 NE independent loads in
    // flight per thread), the slices added one after the other: 0, 1, 2, ...
    constexpr int NE = TI * TJ / 4 / 256;
    const float *src[NE];
    float4 acc[NE];
    bool live[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int q = e * 256 + threadIdx.x, i = i0 + q / (TJ / 4), j = j0 + 4 * (q % (TJ / 4));
      live[e] = i < I && j < J;
      src[e] = a.partial + (live[e] ? (long long)i * J + j : 0);
      acc[e] = make_float4(ld_device_scope(src[e]), ld_device_scope(src[e] + 1), ld_device_scope(src[e] + 2),
                           ld_device_scope(src[e] + 3));
    }
    for (int p = 1; p < a.nsplit; ++p) {
      float4 v[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const float *sp = src[e] + (size_t)p * IJ;
        v[e] = make_float4(ld_device_scope(sp), ld_device_scope(sp + 1), ld_device_scope(sp + 2), ld_device_scope(sp + 3));
      }
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        acc[e].x += v[e].x; acc[e].y += v[e].y; acc[e].z += v[e].z; acc[e].w += v[e].w;
      }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (!live[e]) continue;
      const int q = e * 256 + threadIdx.x, i = i0 + q / (TJ / 4), j = j0 + 4 * (q % (TJ / 4));
      const long long at = i * o.si + out_col(o, j);
      float4 r = acc[e];
      if (o.ep_scale) {
        const float sc = o.ep_scale[i], sh = o.ep_shift[i];
        r.x = __builtin_fmaf(r.x, sc, sh); r.y = __builtin_fmaf(r.y, sc, sh);
        r.z = __builtin_fmaf(r.z, sc, sh); r.w = __builtin_fmaf(r.w, sc, sh);
      }
      if (o.ep_res) {
        const float4 t = *reinterpret_cast<const float4 *>(o.ep_res + at);
        r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
      }
      if (o.ep_relu) {
        r.x = r.x > 0.f ? r.x : 0.f; r.y = r.y > 0.f ? r.y : 0.f;
        r.z = r.z > 0.f ? r.z : 0.f; r.w = r.w > 0.f ? r.w : 0.f;
      }
      *reinterpret_cast<float4 *>(o.D + at) = r;
    }
  } else {
    for (int q = threadIdx.x; q < TI * TJ; q += 256) {
      const int i = i0 + q / TJ, j = j0 + q % TJ;
      if (i >= I || j >= J) continue;
      const float *src = a.partial + (long long)i * J + j;
      float acc = ld_device_scope(src);
      for (int p = 1; p < a.nsplit; ++p) acc += ld_device_scope(src + (size_t)p * IJ);
      out_store(o, i, out_col(o, j), acc);
    }
  }
}

#ifndef CL3D_GEMM_XCD
#define CL3D_GEMM_XCD 1  // (0: the plain tile order, the A/B arm of scripts/micro/kernel_variants.py)
#endif
template <int PREC, int WI, int WJ, int AM, int BM>
__global__ __launch_bounds__(256, 2) void mfma_gemm_kernel(GemmArgs a) {
  constexpr int TI = 64 * WI, TJ = 64 * WJ;
  constexpr int KC = gemm_kc(PREC, WI, WJ);
  using SA = StagePick<PREC, TI, KC>;
  using SB = StagePick<PREC, TJ, KC>;
  // LDS: two copies of the (A, B) chunk pair -- chunk g+1 is written while chunk g is multiplied, one barrier per
  // chunk -- except for the 128 x 128 tile, whose pair is 33 KB (static LDS ends at 64 KB): one copy, two barriers.
  constexpr int NBUF = (WI * WJ == 4) ? 1 : 2;
  constexpr int kPair = SA::kLdsBytes + SB::kLdsBytes;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * kPair];

  // tile / K slice of this workgroup.  Workgroups go to the eight XCDs round robin (blockIdx % 8) and each XCD has its
  // own L2, so the workgroups that read the same operand strips are given to ONE XCD back to back: XCD x takes the x-th
  // eighth of the (slice, tile) list -- slices outermost, the short side of the output fastest (the tail that does not
  // fill a round of eight keeps the plain order).  Neighbours in that list share their strips: the 2-3 channel tiles
  // over one run of points (a layer's product over 65 536 points), the 4-9 tiles of one slice of an early layer's weight
  // gradient; and an XCD's L2 is filled with an eighth of the operand bytes instead of a strip of every slice.
  // Measured on the replayed backbones against the plain order (sessions 14a-c, alternating runs, profiles/r06/):
  // config 2 bf16 6.05 -> 5.95 ms, config 5 20.02 -> 19.91, configs 3 / 4 unchanged, config 2 f32 7.81 -> 7.83; early-stage
  // bf16 products alone 33 -> 28 us (forward, d x), 45 -> 41 (d W).  Two narrower forms were measured too: the list of a
  // slice only (config 2 bf16 6.02), that for >= 16 tiles per slice and the whole list below (6.01).
  int bid = blockIdx.x;
#if CL3D_GEMM_XCD
  {
    const int eighth = (int)gridDim.x >> 3;
    if (bid < (eighth << 3)) bid = (bid & 7) * eighth + (bid >> 3);
  }
#endif
  const int per_slice = a.tiles_i * a.tiles_j;
  const int z = bid / per_slice;  // K slice
  bid -= z * per_slice;
  int ti, tj;
  if (a.tiles_j <= a.tiles_i) {
    tj = bid % a.tiles_j;
    ti = bid / a.tiles_j;
  } else {
    ti = bid % a.tiles_i;
    tj = bid / a.tiles_i;
  }
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

  // Up to two K chunks are kept in flight in registers (sets 0 and 1) on top of the one in LDS, so a chunk's loads
  // have two multiply phases to land even with one wave per SIMD (the widest register footprints -- bf16 above
  // 128 x 128, the scalar fallback -- keep a single set).
  constexpr int DEPTH = ((PREC == PREC_BF16 && WI * WJ >= 2) || AM == STAGE_SCALAR) ? 1 : 2;
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
  auto multiply = [&](int buf, int g) {
    const unsigned char *ldsA = lds + (NBUF == 2 ? buf : 0) * kPair, *ldsB = ldsA + SA::kLdsBytes;
    const int kleft = a.K - g * KC;  // the last chunk of a K that is not a multiple of KC stops early
    if constexpr (PREC == PREC_F32) {
      const float *TA = reinterpret_cast<const float *>(ldsA), *TB = reinterpret_cast<const float *>(ldsB);
      const int strA = SA::type::stride(a.A), strB = SB::type::stride(a.B);
      auto kstep = [&](int kk) {
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
      };
      if (kleft >= KC) {
#pragma unroll 4
        for (int kk = 0; kk < KC / 2; ++kk) kstep(kk);
      } else {
        const int steps = (kleft + 1) / 2;
#pragma unroll 1
        for (int kk = 0; kk < steps; ++kk) kstep(kk);
      }
    } else {
      const uint4 *TA = reinterpret_cast<const uint4 *>(ldsA), *TB = reinterpret_cast<const uint4 *>(ldsB);
      const int steps = kleft >= KC ? KC / 16 : (kleft + 15) / 16;
      for (int ks = 0; ks < steps; ++ks) {
        uint4 fa[WI], fb[WJ];
#pragma unroll
        for (int x = 0; x < WI; ++x) fa[x] = TA[(2 * ks + lh) * (TI + 1) + bf16_swz(wi0 + 32 * x + lr)];
#pragma unroll
        for (int y = 0; y < WJ; ++y) fb[y] = TB[(2 * ks + lh) * (TJ + 1) + bf16_swz(wj0 + 32 * y + lr)];
#pragma unroll
        for (int x = 0; x < WI; ++x)
#pragma unroll
          for (int y = 0; y < WJ; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8 *>(&fa[x]),
                                                                *reinterpret_cast<bf16x8 *>(&fb[y]), acc[x][y], 0, 0, 0);
      }
    }
  };
  auto to_lds = [&](int set, int buf) {
    unsigned char *ldsA = lds + (NBUF == 2 ? buf : 0) * kPair, *ldsB = ldsA + SA::kLdsBytes;
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
  };
  if (g0 < g1) issue(0, g0);
  if constexpr (NBUF == 2) {
    // chunk g sits in LDS copy (g - g0) & 1; register set s carries the chunk that goes to LDS copy s next
    if (DEPTH == 2 && g0 + 1 < g1) issue(1, g0 + 1);
    if (g0 < g1) to_lds(0, 0);
    __syncthreads();
    for (int g = g0; g < g1; g += 2) {
      // --- chunk g out of copy 0
      if constexpr (DEPTH == 2) {
        if (g + 2 < g1) issue(0, g + 2);
      } else {
        if (g + 1 < g1) issue(0, g + 1);
      }
      multiply(0, g);
      if (g + 1 < g1) to_lds(DEPTH == 2 ? 1 : 0, 1);
      __syncthreads();
      if (g + 1 >= g1) break;
      // --- chunk g + 1 out of copy 1
      if constexpr (DEPTH == 2) {
        if (g + 3 < g1) issue(1, g + 3);
      } else {
        if (g + 2 < g1) issue(0, g + 2);
      }
      multiply(1, g + 1);
      if (g + 2 < g1) to_lds(0, 0);
      __syncthreads();
    }
  } else if constexpr (DEPTH == 2) {
    if (g0 + 1 < g1) issue(1, g0 + 1);
    for (int g = g0; g < g1; g += 2) {
      __syncthreads();  // the previous chunk has been multiplied out of LDS
      to_lds(0, 0);
      __syncthreads();
      if (g + 2 < g1) issue(0, g + 2);
      multiply(0, g);
      if (g + 1 < g1) {
        __syncthreads();
        to_lds(1, 0);
        __syncthreads();
        if (g + 3 < g1) issue(1, g + 3);
        multiply(0, g + 1);
      }
    }
  } else {
    for (int g = g0; g < g1; ++g) {
      __syncthreads();
      to_lds(0, 0);
      __syncthreads();
      if (g + 1 < g1) issue(0, g + 1);
      multiply(0, g);
    }
  }

  // D: a lane holds column j = lane & 31 and rows (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) of each 32x32 block
  const int I = a.A.R, J = a.B.R;
  const bool part = a.nsplit > 1;
  float *P = part ? a.partial + (long long)z * I * J : nullptr;
  // kind 0: the result through the output map; 1: this slice's partial tile (summed by a later launch: plain stores, the
  // kernel boundary makes them visible); 2: the same for the in-launch sum -- device-scope stores (written through this
  // XCD's L2: another XCD's workgroup will read them inside this launch, and a fence per workgroup -- an L2 write-back and
  // invalidate each -- serialises on the L2: 63 -> 180 us for 288 .. 2304 workgroups, profiles/r06/gemm_plan_sweep_*fence*)
  auto write_tiles = [&](auto kind) {
#pragma unroll
    for (int x = 0; x < WI; ++x)
#pragma unroll
      for (int y = 0; y < WJ; ++y) {
        const int j = j0 + wj0 + 32 * y + lr;
        const long long joff = decltype(kind)::value ? j : out_col(a.out, j);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = i0 + wi0 + 32 * x + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (i < I && j < J) {
            if constexpr (decltype(kind)::value == 2)
              __hip_atomic_store(P + (long long)i * J + j, acc[x][y][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if constexpr (decltype(kind)::value == 1) P[(long long)i * J + j] = acc[x][y][e];
            else out_store(a.out, i, joff, acc[x][y][e]);
          }
        }
      }
  };
  if (!part) write_tiles(std::integral_constant<int, 0>{});
  else if (a.tickets == nullptr) write_tiles(std::integral_constant<int, 1>{});
  else {
    write_tiles(std::integral_constant<int, 2>{});
    sum_slices_by_last_arrival<TI, TJ>(a, i0, j0, ti * a.tiles_j + tj);
  }
}

// ---- slice-ordered sum of the split-K partials.  A workgroup owns 16 consecutive elements of the I x J tile; its 16
// thread groups take the slices s = g, g+16, g+32, ... and the 16 sub-sums are added in group order through LDS: the
// summation order depends on nsplit only (bit-reproducible).  MODE 0 sends element (i, j) through the output map
// (strides, folded clouds, inference epilogue); MODE 1 turns d wcat [2Co, C] (+ d W_r) into d W [Co, 3+2C].
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float *__restrict__ part, int nsplit, int I, int J,
                                                          OutMap o, const float *__restrict__ dwr, int Co, int C) {
  // VEC = 4: a thread owns four consecutive elements (16-byte loads of every slice; the caller guarantees that four
  // neighbours never straddle a row: J % 4 == 0 resp. C % 4 == 0) -- the 512-slice weight-gradient reduce of the
  // metric shape went 7.5 -> ~4 us
  __shared__ float s_sum[2][16][16 * VEC + 1];
  const int el = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const long long IJ = (long long)I * J;
  const long long n_el = MODE == 0 ? IJ : (long long)Co * C;
  for (long long e0 = (long long)blockIdx.x * 16 * VEC; e0 < n_el; e0 += (long long)gridDim.x * 16 * VEC) {
    const long long e = e0 + (long long)el * VEC;
    float top[VEC], bot[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) top[v] = bot[v] = 0.f;
    if (e < n_el) {
#pragma unroll 4
      for (int p = sg; p < nsplit; p += 16) {  // (unrolled: four slices' loads in flight, added in the same order)
        if constexpr (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4 *>(part + (size_t)p * IJ + e);
          top[0] += t.x; top[1] += t.y; top[2] += t.z; top[3] += t.w;
          if (MODE == 1) {
            const float4 u = *reinterpret_cast<const float4 *>(part + (size_t)p * IJ + (size_t)Co * C + e);
            bot[0] += u.x; bot[1] += u.y; bot[2] += u.z; bot[3] += u.w;
          }
        } else {
          top[0] += part[(size_t)p * IJ + e];
          if (MODE == 1) bot[0] += part[(size_t)p * IJ + (size_t)Co * C + e];  // bot = d wcat[Co + o][c]
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      s_sum[0][sg][el * VEC + v] = top[v];
      s_sum[1][sg][el * VEC + v] = bot[v];
    }
    __syncthreads();
    if (sg == 0 && e < n_el) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float t = s_sum[0][0][el * VEC + v], bt = s_sum[1][0][el * VEC + v];
#pragma unroll
        for (int g = 1; g < 16; ++g) {
          t += s_sum[0][g][el * VEC + v];
          bt += s_sum[1][g][el * VEC + v];
        }
        const long long ev = e + v;
        if (MODE == 0) {
          const int i = (int)(ev / J), j = (int)(ev - (long long)i * J);
          out_store(o, i, out_col(o, j), t);
        } else {
          const int oo = (int)(ev / C), c = (int)(ev - (long long)oo * C);
          const int ld = 3 + 2 * C;
          o.D[(size_t)oo * ld + 3 + c] = bt;            // d W_c
          o.D[(size_t)oo * ld + 3 + C + c] = t - bt;    // d W_d
          if (c < 3) o.D[(size_t)oo * ld + c] = dwr ? dwr[oo * 3 + c] : 0.f;
        }
      }
    }
    __syncthreads();
  }
}

// few slices (the forward / input-gradient products: nsplit <= 16): one thread per four consecutive elements of a row,
// slices added in order p = 0, 1, ...; 16-byte loads and stores, no LDS.  Needs J % 4 == 0 and, for a folded output,
// fold_n % 4 == 0 (four neighbours never straddle a row or a cloud).
__global__ __launch_bounds__(256) void gemm_reduce4_kernel(const float *__restrict__ part, int nsplit, int I, int J, OutMap o) {
  const long long IJ = (long long)I * J, n4 = IJ / 4;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
    const long long e = q * 4;
    float4 acc = *reinterpret_cast<const float4 *>(part + e);
#pragma unroll 4
    for (int p = 1; p < nsplit; ++p) {
      const float4 v = *reinterpret_cast<const float4 *>(part + (size_t)p * IJ + e);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const int i = (int)(e / J), j = (int)(e - (long long)i * J);
    const long long at = i * o.si + out_col(o, j);
    if (o.ep_scale) {
      const float sc = o.ep_scale[i], sh = o.ep_shift[i];
      acc.x = __builtin_fmaf(acc.x, sc, sh); acc.y = __builtin_fmaf(acc.y, sc, sh);
      acc.z = __builtin_fmaf(acc.z, sc, sh); acc.w = __builtin_fmaf(acc.w, sc, sh);
    }
    if (o.ep_res) {
      const float4 r = *reinterpret_cast<const float4 *>(o.ep_res + at);
      acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
    }
    if (o.ep_relu) {
      acc.x = acc.x > 0.f ? acc.x : 0.f; acc.y = acc.y > 0.f ? acc.y : 0.f;
      acc.z = acc.z > 0.f ? acc.z : 0.f; acc.w = acc.w > 0.f ? acc.w : 0.f;
    }
    *reinterpret_cast<float4 *>(o.D + at) = acc;
  }
}

// MODE 1 of gemm_reduce_kernel for few slices (nsplit <= 16: the deep layers' weight gradients, K = 256 .. 4096 points): one
// thread per four consecutive input channels of an output row -- d wcat's two halves read with 16-byte loads, slices added in
// order p = 0, 1, ... (the order gemm_reduce_kernel's sixteen slice groups give for nsplit <= 16), every thread storing.
// gemm_reduce_kernel<1, 4> spends a 256-thread workgroup on 64 elements with nsplit x 16 threads loading and 16 storing:
// 34 us for the 1152 x 1152 layer's 10.6 MB, 22 us for 576 x 576 (profiles/r06/session26_summary.txt).  Needs C % 4 == 0.
__global__ __launch_bounds__(256) void gemm_reduce_w4_kernel(const float *__restrict__ part, int nsplit, long long IJ,
                                                             float *__restrict__ dW, const float *__restrict__ dwr, int Co,
                                                             int C) {
  const long long half = (long long)Co * C, n4 = half / 4;
  const int ld = 3 + 2 * C;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
    const long long e = q * 4;
    float4 t = *reinterpret_cast<const float4 *>(part + e);         // d wcat[o][c]      = d (W_d) part
    float4 b = *reinterpret_cast<const float4 *>(part + half + e);  // d wcat[Co + o][c] = d (W_c - W_d) part
#pragma unroll 4
    for (int p = 1; p < nsplit; ++p) {
      const float4 u = *reinterpret_cast<const float4 *>(part + (size_t)p * IJ + e);
      const float4 v = *reinterpret_cast<const float4 *>(part + (size_t)p * IJ + half + e);
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
    const int oo = (int)(e / C), c = (int)(e - (long long)oo * C);
    float *row = dW + (size_t)oo * ld;
    row[3 + c] = b.x; row[3 + c + 1] = b.y; row[3 + c + 2] = b.z; row[3 + c + 3] = b.w;                  // d W_c
    row[3 + C + c] = t.x - b.x; row[3 + C + c + 1] = t.y - b.y; row[3 + C + c + 2] = t.z - b.z;          // d W_d
    row[3 + C + c + 3] = t.w - b.w;
    if (c == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) row[k] = dwr ? dwr[oo * 3 + k] : 0.f;
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

// ---- the PointWiseMLP's forward per-point product WITHOUT LDS and on <= 96 VGPRs ---------------------------------
// ght[p][j] = sum_c act(F[c][p]) wcat[j][c] for channel-major F and few channels (C <= 72).  In the training step this
// product runs beside the ball query, whose one workgroup per CU holds ~147 KB of LDS and 4 x 104 VGPRs per SIMD lane:
// mfma_gemm_kernel (33 KB of LDS per workgroup) cannot be resident next to it and ran in the ball query's tail -- the
// statistics pass, which needs both, started ~25 us after the ball query had finished.  This form needs no LDS at all
// and fits the registers the ball query leaves: a wave owns one 32-column tile of the output (its wcat fragments, C / 2
// VGPRs, loaded once) and walks 32-point blocks; a block's operand fragments come STRAIGHT from global memory (32
// consecutive points of one channel are 128 contiguous bytes = the lanes of one v_mfma_f32_32x32x2_f32 fragment; the
// four waves of a workgroup = four column tiles re-read the block out of L1).  Same instruction, same f32 accumulation as
// the staged kernel; the k order of the sum is the same too (ascending), so are the results.
template <int CH>  // CH = ceil(C / 2) MFMA steps
__global__ __launch_bounds__(256, 5) void pwmlp_rows_nolds_kernel(const float *__restrict__ F, const float *__restrict__ pro_scale,
                                                                  const float *__restrict__ pro_shift,
                                                                  const float *__restrict__ wcat, float *__restrict__ out,
                                                                  int C, int N, int J, int nblocks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int j = 32 * (blockIdx.y * 4 + wave) + lr;  // this lane's output column
  if (32 * ((int)blockIdx.y * 4 + wave) >= J) return;  // (whole waves: no barrier in this kernel)
  float bw[CH];
#pragma unroll
  for (int s = 0; s < CH; ++s) {
    const int k = 2 * s + lh;
    bw[s] = (j < J && k < C) ? wcat[(size_t)j * C + k] : 0.f;
  }
  const int per_cloud = N / 32;
  // wave-uniform bases + ONE 32-bit lane offset per side (saddr + voffset accesses): a 64-bit address per fragment load
  // and per store would alone be more registers than the kernel may have
  const unsigned a_off = ((unsigned)lh * (unsigned)N + (unsigned)lr) * 4u;
  const unsigned o_off = ((unsigned)(4 * lh) * (unsigned)J + (unsigned)(j < J ? j : 0)) * 4u;
  const bool odd_tail = (C & 1) != 0 && lh == 1;  // the last step's second k does not exist
  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int b = __builtin_amdgcn_readfirstlane(blk / per_cloud);
    const int n0 = __builtin_amdgcn_readfirstlane((blk - b * per_cloud) * 32);
    const char *fb = reinterpret_cast<const char *>(F + (size_t)b * C * N + n0);
    float a[CH];
#pragma unroll
    for (int s = 0; s < CH; ++s) {
      const char *row = fb + (size_t)(2 * s) * N * 4u;  // uniform
      a[s] = (s == CH - 1 && odd_tail) ? 0.f : *reinterpret_cast<const float *>(row + a_off);
    }
    if (pro_scale != nullptr) {  // the producing layer's BatchNorm + ReLU, applied to the fragment
#pragma unroll
      for (int s = 0; s < CH; ++s) {
        const bool dead = s == CH - 1 && odd_tail;
        const int k = dead ? 0 : 2 * s + lh;
        const float z = __builtin_fmaf(a[s], pro_scale[k], pro_shift[k]);
        a[s] = dead ? 0.f : (z > 0.f ? z : 0.f);
      }
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < CH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bw[s], acc, 0, 0, 0);
    if (j < J) {
      char *ob = reinterpret_cast<char *>(out + ((size_t)b * N + n0) * J);
#pragma unroll
      for (int e = 0; e < 16; ++e)
        *reinterpret_cast<float *>(ob + (size_t)((e & 3) + 8 * (e >> 2)) * J * 4u + o_off) = acc[e];
    }
  }
}

// launches it when the shape qualifies (f32, C in {32, 36, 64, 72}: the layers that run beside a ball query at full
// resolution, whole 32-point blocks per cloud); false = the caller takes the staged kernel
static bool launch_rows_nolds(const float *F, const float *pro_scale, const float *pro_shift, const float *wcat, float *ght,
                              int B, int C, int N, int Co, hipStream_t st) {
  if ((N & 31) != 0 || B < 1) return false;
  const int J = 2 * Co;
  const long long nblocks = (long long)B * (N / 32);
  if (nblocks > 0x7fffffffLL) return false;
  const dim3 grid((unsigned)(nblocks < 256 ? nblocks : 256), (unsigned)ceil_div(ceil_div(J, 32), 4));  // one workgroup per CU
#define CL3D_NOLDS(CH_) \
  hipLaunchKernelGGL((pwmlp_rows_nolds_kernel<CH_>), grid, dim3(256), 0, st, F, pro_scale, pro_shift, wcat, ght, C, N, J, (int)nblocks)
  switch (C) {
    case 32: CL3D_NOLDS(16); return true;
    case 36: CL3D_NOLDS(18); return true;
    case 64: CL3D_NOLDS(32); return true;
    case 72: CL3D_NOLDS(36); return true;
    default: return false;
  }
#undef CL3D_NOLDS
}

// the same product with bf16 inputs (round 4; prototype and stand-alone check: scripts/micro/skinny_gemm_bf16.hip, 14 us
// at the metric shape against 22.8 us for the staged bf16 kernel, which moreover cannot run beside the ball query): a
// lane's A fragment for one v_mfma_f32_32x32x16_bf16 step is 8 channels of ONE point -- eight 4-byte loads, each a
// 128-byte run across the 32 lanes of a half-wave -- rounded to bf16 (RNE, after the producing layer's BatchNorm + ReLU
// when there is one) and packed; the weight fragments (C / 16 packs of 8 bf16) stay in registers.  Same number of
// loads per block as the f32 kernel, an eighth of its MFMA cycles.
template <int C, bool PRO>  // channels (compile time: the dead channels of a last step fold away), producer prologue
__global__ __launch_bounds__(256, 5) void pwmlp_rows_nolds_bf16_kernel(const float *__restrict__ F, const float *__restrict__ pro_scale,
                                                                       const float *__restrict__ pro_shift,
                                                                       const float *__restrict__ wcat, float *__restrict__ out,
                                                                       int N, int J, int nblocks) {
  constexpr int CH16 = (C + 15) / 16;  // MFMA steps
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int j = 32 * (blockIdx.y * 4 + wave) + lr;  // this lane's output column
  if (32 * ((int)blockIdx.y * 4 + wave) >= J) return;  // (whole waves: no barrier in this kernel)
  bf16x8 bw[CH16];
#pragma unroll
  for (int s = 0; s < CH16; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * s + 8 * lh + e;
      bw[s][e] = (__bf16)((j < J && k < C) ? wcat[(size_t)j * C + k] : 0.f);
    }
  const int per_cloud = N / 32;
  const unsigned a_off = ((unsigned)(8 * lh) * (unsigned)N + (unsigned)lr) * 4u;  // lane part of every fragment address
  const unsigned o_off = ((unsigned)(4 * lh) * (unsigned)J + (unsigned)(j < J ? j : 0)) * 4u;
  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int b = __builtin_amdgcn_readfirstlane(blk / per_cloud);
    const int n0 = __builtin_amdgcn_readfirstlane((blk - b * per_cloud) * 32);
    const char *fb = reinterpret_cast<const char *>(F + (size_t)b * C * N + n0);
    bf16x8 a[CH16];
#pragma unroll
    for (int s = 0; s < CH16; ++s) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool live = 16 * s + 8 * lh + e < C;  // (only the last step of C % 16 != 0 has dead channels)
        x[e] = live ? *reinterpret_cast<const float *>(fb + (size_t)(16 * s + e) * N * 4u + a_off) : 0.f;
      }
      if constexpr (PRO) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 16 * s + 8 * lh + e;
          const bool live = k < C;
          // (the index goes through an empty asm: left to itself the compiler keeps all 2 C / 16 x 8 per-lane constants
          // of the prologue in registers across the block loop and spills; they are L1 hits)
          int kk = live ? k : 0;
          asm volatile("" : "+v"(kk));
          const float z = __builtin_fmaf(x[e], pro_scale[kk], pro_shift[kk]);
          x[e] = live ? (z > 0.f ? z : 0.f) : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) a[s][e] = (__bf16)x[e];
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < CH16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], bw[s], acc, 0, 0, 0);
    if (j < J) {
      char *ob = reinterpret_cast<char *>(out + ((size_t)b * N + n0) * J);
#pragma unroll
      for (int e = 0; e < 16; ++e)
        *reinterpret_cast<float *>(ob + (size_t)((e & 3) + 8 * (e >> 2)) * J * 4u + o_off) = acc[e];
    }
  }
}

static bool launch_rows_nolds_bf16(const float *F, const float *pro_scale, const float *pro_shift, const float *wcat, float *ght,
                                   int B, int C, int N, int Co, hipStream_t st) {
  if ((N & 31) != 0 || B < 1) return false;
  const int J = 2 * Co;
  const long long nblocks = (long long)B * (N / 32);
  if (nblocks > 0x7fffffffLL) return false;
  const dim3 grid((unsigned)(nblocks < 256 ? nblocks : 256), (unsigned)ceil_div(ceil_div(J, 32), 4));
#define CL3D_NOLDS16(C_)                                                                                                        \
  do {                                                                                                                          \
    if (pro_scale != nullptr)                                                                                                   \
      hipLaunchKernelGGL((pwmlp_rows_nolds_bf16_kernel<C_, true>), grid, dim3(256), 0, st, F, pro_scale, pro_shift, wcat, ght, N, J, \
                         (int)nblocks);                                                                                         \
    else                                                                                                                        \
      hipLaunchKernelGGL((pwmlp_rows_nolds_bf16_kernel<C_, false>), grid, dim3(256), 0, st, F, pro_scale, pro_shift, wcat, ght, N, J, \
                         (int)nblocks);                                                                                         \
  } while (0)
  switch (C) {
    case 32: CL3D_NOLDS16(32); return true;
    case 36: CL3D_NOLDS16(36); return true;
    case 64: CL3D_NOLDS16(64); return true;
    case 72:
      if (pro_scale != nullptr) return false;  // (the prologue form spills at 72 channels: the staged kernel takes it)
      CL3D_NOLDS16(72);
      return true;
    default: return false;
  }
#undef CL3D_NOLDS16
}

// ---- host side -------------------------------------------------------------------------------------------------
// ---- BOTH gradients of the PointWiseMLP's per-point product from ONE pass over d ght (round 5) -------------------------
//   d F[c][p]    = sum_o wcat[o][c] dght[p][o]      (data gradient, channel-major out)
//   d wcat[o][c] = sum_p dght[p][o] F[c][p]         (weight gradient, over every point of the batch)
// Round 4 ran them as two launches of mfma_gemm_kernel side by side on two queues (25 + 37 us alone, 41 + 48 together,
// + 9 us for the 512-slice reduce): each read the 33.5 MB of d ght, the weight gradient as 512 K slices of a 128 x 64
// tile whose partials were written and read again.  Here a workgroup owns a run of 64-point tiles: a tile's d ght
// rows (32 KB) and feature rows (16 KB) are staged ONCE in LDS and feed both products; the weight-gradient
// accumulators stay in registers across the workgroup's tiles (256 partial tiles instead of 512, written once);
// wcat, the data gradient's constant operand, lives in registers for the whole launch.
//   LDS     T [64 points][129]: d ght rows, row stride odd, so BOTH fragment orientations are conflict-free single reads:
//             data gradient    B(k = o, j = p)   lane (p = lr, o = 2kk + lh)  -> address p*129 + o: 32 banks over lr
//             weight gradient  A(i = o, k = p)   lane (o = 32w + lr, p = 2kk + lh) -> address p*129 + o: consecutive
//           Fs[64 channels][65]: feature rows; weight gradient B(k = p, j = c) lane (c = lr, p = 2kk + lh) -> c*65 + p
//           two copies of the pair: tile t+1 is written while tile t is multiplied, one barrier per tile.
//   waves   data gradient 64 channels x 64 points = 2 x 2 blocks of 32 x 32, one per wave, 64 MFMA steps over o;
//           weight gradient 128 x 64 = 4 x 2 blocks, wave w owns rows o = 32w .. 32w+31, 32 steps over the tile's points.
//   order   sums run over points in ascending order inside a workgroup; the workgroups' partial tiles are added in
//           workgroup order by gemm_reduce_kernel<1> (which also does the weight plumbing): bit-reproducible, and
//           independent of the device (the grid is min(256, tiles), not the CU count).
// C <= 64 channels and 2 Co <= 128 columns (zero padding below that), whole 64-point tiles per cloud.  The optional
// prologue is the bottleneck's conv1 BatchNorm + ReLU applied to the feature rows as they are staged.
struct PointGradArgs {
  const float *dght, *F, *pro_scale, *pro_shift, *wcat;
  float *dfeat, *partial;
  int C, J, N, tiles, tiles_per_cloud;
};

constexpr int kPgTP = 64, kPgLdT = 129, kPgLdF = 65;
constexpr int kPgBufFloats = kPgTP * kPgLdT + 64 * kPgLdF;
constexpr int kPointGradsGrid = 256;

// FULL: C == 64 and J == 128 (no padding, no masks: straight-line staging code, exact s_waitcnt counts -- with the masked
// loads in the loop the compiler waited for EVERY outstanding load, the next tile's included, before the first MFMA);
// PRO: the feature rows enter as max(scale[c] x + shift[c], 0)
template <bool FULL, bool PRO>
__global__ __launch_bounds__(256) void pwmlp_point_grads_kernel(PointGradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float pg_lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int ci = wave >> 1, pi = wave & 1;  // this wave's block of the data gradient
  const int C = FULL ? 64 : a.C, J = FULL ? 128 : a.J, N = a.N;

  // the data gradient's A fragments: wcat[o = 2kk + lh][c = 32ci + lr], constant for the launch
  float wreg[64];
  {
    const int c = 32 * ci + lr;
#pragma unroll
    for (int kk = 0; kk < 64; ++kk) {
      const int o = 2 * kk + lh;
      wreg[kk] = (FULL || (o < J && c < C)) ? a.wcat[(size_t)o * C + c] : 0.f;
    }
  }
  f32x16 acc2[2];
#pragma unroll
  for (int y = 0; y < 2; ++y)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[y][e] = 0.f;

  // staging: d ght rows as 32 lanes x 16 bytes per row (8 rows per pass of the workgroup), feature rows as 16 lanes x
  // 16 bytes per channel (16 channels per pass)
  const int tr = t >> 5, tc = 4 * (t & 31);
  const int fc = t >> 4, fp = 4 * (t & 15);
  float4 gT[8], gF[4];
  auto issue = [&](int tile) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int b = tile / a.tiles_per_cloud, n0 = (tile - b * a.tiles_per_cloud) * kPgTP;
    const float *src = a.dght + ((size_t)b * N + n0) * J + tc;
#pragma unroll
    for (int q = 0; q < 8; ++q) gT[q] = (FULL || tc < J) ? ld4(src + (size_t)(tr + 8 * q) * J) : zero4;
    const float *fs = a.F + (size_t)b * C * N + n0 + fp;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = fc + 16 * q;
      float4 v = (FULL || c < C) ? ld4(fs + (size_t)c * N) : zero4;
      if (PRO && (FULL || c < C)) {
        const float sc = a.pro_scale[c], sh = a.pro_shift[c];
        v.x = __builtin_fmaf(v.x, sc, sh); v.y = __builtin_fmaf(v.y, sc, sh);
        v.z = __builtin_fmaf(v.z, sc, sh); v.w = __builtin_fmaf(v.w, sc, sh);
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
      }
      gF[q] = v;
    }
  };
  auto to_lds = [&](int buf) {
    float *T = pg_lds + buf * kPgBufFloats, *Fs = T + kPgTP * kPgLdT;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float *d = T + (tr + 8 * q) * kPgLdT + tc;
      d[0] = gT[q].x; d[1] = gT[q].y; d[2] = gT[q].z; d[3] = gT[q].w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float *d = Fs + (fc + 16 * q) * kPgLdF + fp;
      d[0] = gF[q].x; d[1] = gF[q].y; d[2] = gF[q].z; d[3] = gF[q].w;
    }
  };

  int tile = blockIdx.x, buf = 0;
  if (tile < a.tiles) {
    issue(tile);
    to_lds(0);
  }
  // the wcat fragments are pinned as ARRIVED here: left to itself the compiler sinks their loads below the barrier, and
  // the tile loop's s_waitcnt logic then has to assume some are still outstanding on every iteration -- it waited for
  // the NEXT tile's loads (issued at the top of the iteration) before the first MFMA, i.e. no prefetch at all
#pragma unroll
  for (int kk = 0; kk < 64; ++kk) asm volatile("" : "+v"(wreg[kk]));
  __syncthreads();
  for (; tile < a.tiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < a.tiles) issue(next);  // in flight while this tile is multiplied
    const float *T = pg_lds + buf * kPgBufFloats, *Fs = T + kPgTP * kPgLdT;
    if (a.dfeat != nullptr) {
      f32x16 acc1;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[e] = 0.f;
      const float *tb = T + (32 * pi + lr) * kPgLdT + lh;
#pragma unroll
      for (int kk = 0; kk < 64; ++kk) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[kk], tb[2 * kk], acc1, 0, 0, 0);
      // a lane holds point 32pi + lr and channels 32ci + (e & 3) + 8 (e >> 2) + 4 lh: 128-byte runs per store
      const int b = tile / a.tiles_per_cloud, n0 = (tile - b * a.tiles_per_cloud) * kPgTP;
      float *ob = a.dfeat + (size_t)b * C * N + n0 + 32 * pi + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int c = 32 * ci + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (FULL || c < C) ob[(size_t)c * N] = acc1[e];
      }
    }
    if (a.partial != nullptr) {
      const float *ta = T + lh * kPgLdT + 32 * wave + lr;
      const float *f0 = Fs + lr * kPgLdF + lh, *f1 = f0 + 32 * kPgLdF;
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) {
        const float av = ta[2 * kk * kPgLdT];
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, f0[2 * kk], acc2[0], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, f1[2 * kk], acc2[1], 0, 0, 0);
      }
    }
    if (next < a.tiles) to_lds(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if (a.partial != nullptr) {
    // this workgroup's partial d wcat [J][C]: a lane holds column c = 32y + lr and rows o = 32w + (e & 3) + 8 (e >> 2) + 4 lh
    float *P = a.partial + (size_t)blockIdx.x * J * C;
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int c = 32 * y + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int o = 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (FULL || (o < J && c < C)) P[(size_t)o * C + c] = acc2[y][e];
      }
    }
  }
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int log2_exact(int n) { return n > 0 && (n & (n - 1)) == 0 ? __builtin_ctz((unsigned)n) : -1; }

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
  s.fold_shift = log2_exact(n);
  if (point_on_r) {
    s.sr = 1; s.sk = n; s.R = nb * n; s.rc = 1; s.fold = nb > 1 ? 1 : 0;
  } else {
    s.sr = n; s.sk = 1; s.R = rows; s.rc = 0; s.fold = nb > 1 ? 2 : 0;
  }
  s.vec = aligned16(p) && n % 4 == 0;
  return s;
}

static int stage_mode(const GemmOperand &s) { return !s.vec ? STAGE_SCALAR : (s.rc ? STAGE_VEC_RC : STAGE_VEC_KC); }

// BatchNorm + ReLU of the producing layer applied while this operand is staged: channel = k (axis 0) or r (axis 1).
// The vector staging paths read four consecutive channels' coefficients at once where the channel axis is the
// contiguous one: without 16-byte aligned coefficient arrays the operand falls back to element-wise staging.
static void set_prologue(GemmOperand &s, const float *scale, const float *shift, int axis) {
  s.pro_scale = scale; s.pro_shift = shift; s.pro_axis = axis;
  const bool channel_contiguous = (axis == 1) == (s.rc != 0);
  if (scale != nullptr && channel_contiguous && !(aligned16(scale) && aligned16(shift))) s.vec = 0;
}

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


// (An LDS-ring variant of this kernel -- stages written by global_load_lds_dwordx4 DMA three chunks ahead, counted
// s_waitcnt vmcnt, one raw s_barrier per chunk, inline-asm fragment reads -- was built and measured in round 2: equal
// results, 5-10 % faster on the deepest layers (K >= 576, 64 points per cloud), 20-35 % SLOWER on the 4096-point
// layers and, with ~100 KiB of LDS per workgroup, it starves whatever kernel runs beside it on another stream
// (the ball query next to the per-point GEMM: 148 us instead of 53).  Removed; see git history and DESIGN.md 3.2.)


constexpr int kCUs = 256;
// Where the in-launch slice sum is used: measured against the two-launch form over every (tile, split) of the config-2
// backbone's 42 products, both precisions (profiles/r06/gemm_plan_sweep_*: the same graph-replayed timing, launches ~1 us
// apart): the last arrival of a tile streams its slices at ~17 KB/us -- ONE workgroup, loads that bypass its L2 -- while the
// launch of its own spreads the same bytes over the chip.  In-launch minus two-launch, mean over the deep-stage products:
//     64 x 64 tile:   split 2: -0.8 .. -1.5 us   3: -0.5 .. -1.5   4: +0.8   8: +6    16: +14 .. +16
//     128 x 64:             2: +0.3 .. +1.0      3: +0.1 .. +1.2   4: +4 .. +5       16: +26 .. +40
//     128 x 128:            2: +4                3: +5 .. +8       8: +28            16: +63 .. +76
// So: up to three slices of a 64 x 64 tile; everything else keeps the launch of its own.
constexpr int kTicketMaxSplit = 3;
constexpr double kInLaunchSumGainUs = 1.0;  // what the in-launch form saves where it is used (see the table above)


// Workgroup tile and K slicing for an I x J output over K.  Two regimes decide the time of a product here:
//   * matrix-core bound: padded flops / 157 TFLOP/s (f32) -- padding matters, these layers have I or J = 72 .. 288;
//   * latency bound: a workgroup keeps <= 2 chunks (16-32 KB) in flight, and ~100 KB per CU are needed to cover
//     HBM/L2 latency at the rate the MFMAs consume operands, so a CU wants `resident` workgroups (what registers and
//     LDS admit) and the grid >= CUs x resident; with fewer the time stretches by that ratio.
// Outputs with few tiles and a long K (the deep stages: 256 points x 2304 channels; every weight gradient) are
// therefore cut along K into slices whose partial tiles are summed in slice order by gemm_reduce_kernel; a slice
// costs its partial tile written and read once (priced at 4 TB/s), keeps >= 4 chunks, and the partials must fit the
// caller's scratch.  The candidate with the smallest estimate wins; ties go to the larger tile.
struct Plan {
  int wi, wj, nsplit, cps;
};

// (variant builds of scripts/micro/gemm_plan_sweep.py only: CL3D_GEMM_FUSED_SUM=0 restores the two-launch slice sum for the A/B)
static bool fused_slice_sum() {
#ifdef CL3D_GEMM_PLAN_ENV
  const char *e = getenv("CL3D_GEMM_FUSED_SUM");  // (read per call: the sweep switches it between timings)
  return !(e && e[0] == '0');
#else
  return true;
#endif
}

// bf16: what a launch costs was measured over every tile and K split on the convolutions of the config-2 backbone
// (scripts/micro/gemm_plan_sweep.py, 14 layers x 3 products x 4 tiles x 8-9 splits, launches replayed from a HIP graph;
// profiles/r05/gemm_plan_sweep_bf16.jsonl) and does not follow the flop count at all -- these products are 1-3 GFLOP and
// 7-75 MB, and a launch is a chain of memory round trips:
//     T = T0 + RT + generations x ((chunks per workgroup - 1) x t_chunk + t_tail)  [+ the slice-sum launch]
// T0 + RT ~ 7.5 us before the first chunk is in LDS; t_chunk = what one more 64-deep chunk costs a workgroup: its two
// register sets cover one round trip between them (1.2 us for the 64 x 64 tile, 1.7 for 128 x 64, 0.65-2 for 128 x 128)
// unless the workgroups resident on a CU together ask for more than the ~36 GB/s a CU draws (then chunk bytes x resident /
// that rate); t_tail = the epilogue and the hand-over to the next workgroup on the CU; generations = workgroups / (256 CUs
// x resident).  Least-squares fit of the logarithm: 9.7 % rms over the 1400 timings.  Plans chosen by the fit total 955 us
// over the 42 products against 906 us for the best plan of each and 1204 us for the flop-count model this replaces (which
// stays for f32, where it is within 5 % of the best: 1739 against 1649 us).
// what summing `split` slices costs on top of the product: a launch of its own over the whole chip (two-launch form), or
// the last arrival of every tile reading its tile of every slice (in-launch form: no launch, no hand-over, but ONE
// workgroup per tile streams split x tile bytes at what a single workgroup draws)
static bool in_launch_sum_applies(int wi, int wj, long long split, bool in_launch) {
  return in_launch && split > 1 && split <= kTicketMaxSplit && wi == 1 && wj == 1;
}
static double slice_sum_us(int I, int J, int wi, int wj, long long split, bool in_launch) {
  if (split <= 1) return 0.0;
  const double two_launch = 1.546 + (double)split * I * J * 4.0 * 2.0 / 1e6 / 6.851;  // (round 5's fit)
  return in_launch_sum_applies(wi, wj, split, in_launch) ? two_launch - kInLaunchSumGainUs : two_launch;
}

static double bf16_launch_us(int I, int J, long long K, int wi, int wj, long long cps, long long real_split, bool in_launch) {
  const int t = (wi == 2 ? 1 : 0) + (wj == 2 ? 2 : 0);  // 64x64, 128x64, 64x128, 128x128
  const double kChunk[4] = {1.171, 1.682, 1.644, 0.65}, kTail[4] = {5.662, 4.682, 3.924, 5.453};
  const double kResident[4] = {4, 3, 3, 2};
  const double T0 = 4.3, RT = 3.178, cu_kb_per_us = 35.636;
  const long long ti = ceil_div(I, 64 * wi), tj = ceil_div(J, 64 * wj);
  const double wgs = (double)(ti * tj * real_split);
  double conc = wgs / kCUs;
  conc = conc < 1.0 ? 1.0 : (conc > kResident[t] ? kResident[t] : conc);
  const double chunk_kb = (64.0 * wi + 64.0 * wj) * 64.0 * 4.0 / 1e3;
  double t_chunk = conc * chunk_kb / cu_kb_per_us;
  if (t_chunk < kChunk[t]) t_chunk = kChunk[t];
  double gens = wgs / (kCUs * kResident[t]);
  if (gens < 1.0) gens = 1.0;
  double us = T0 + RT + gens * ((double)(cps - 1) * t_chunk + kTail[t]);
  return us + slice_sum_us(I, J, wi, wj, real_split, in_launch);
}

static Plan plan_gemm(int I, int J, long long K, int precision, int max_split, size_t ws_bytes, bool no_big_tile = false,
                      bool in_launch_sum = false) {
  const int cand[4][3] = {{2, 2, 2}, {2, 1, 3}, {1, 2, 3}, {1, 1, 4}};  // wi, wj, resident workgroups per CU
  const double peak_flops_per_us = 157.3e6;  // f32-input MFMA
  Plan best{2, 1, 1, (int)((K + gemm_kc(precision, 2, 1) - 1) / gemm_kc(precision, 2, 1))};
  double best_cost = 1e300;
  for (int c = 0; c < 4; ++c) {
    const int wi = cand[c][0], wj = cand[c][1], resident = cand[c][2];
    // 128 x 128 is left out where it does not fit the register file: the element-wise staging fallback (144 spilled
    // registers) and, in f32, the pair of k-contiguous operands (a convolution's weight gradient: 38).  A kernel that
    // spills needs the queue's private scratch, and two such launches on concurrent branches of a HIP graph gave wrong
    // results on this stack -- no kernel this planner can pick spills.
    if (no_big_tile && wi * wj == 4) continue;
    const int kc = gemm_kc(precision, wi, wj);
    const long long chunks = (K + kc - 1) / kc;
    const long long ti = ceil_div(I, 64 * wi), tj = ceil_div(J, 64 * wj), tiles = ti * tj;
    const double flops = 2.0 * (double)(ti * 64 * wi) * (double)(tj * 64 * wj) * (double)K;
    // f32 keeps >= 4 chunks per slice; bf16 slices go down to 2 (the fit above prices what that costs)
    const long long min_cps = precision == PREC_BF16 ? 2 : 4;
    for (long long split = 1; split <= max_split && split <= (chunks >= min_cps ? chunks / min_cps : 1); split += (split < 4 ? 1 : split / 2)) {
      if (split > 1 && (size_t)split * I * J * sizeof(float) > ws_bytes) break;
      const long long cps = (chunks + split - 1) / split;
      const long long real_split = (chunks + cps - 1) / cps;
      double cost;
      if (precision == PREC_BF16) {
        cost = bf16_launch_us(I, J, K, wi, wj, cps, real_split, in_launch_sum);
      } else {
        double fill = (double)(tiles * real_split) / (double)(kCUs * resident);
        if (fill > 1.0) fill = 1.0;
        const double partial_us = real_split <= 1 ? 0.0
                                  : 3.0 + 2.0 * (double)real_split * I * J * 4.0 / 2.5e6 -
                                        (in_launch_sum_applies(wi, wj, real_split, in_launch_sum) ? kInLaunchSumGainUs : 0.0);
        // (round 5: a term for the operand bytes a tiling pulls through L2 -- tiles x K x (TI + TJ) x 4 B against 3-6 TB/s,
        //  170 MB for the 1152 x 1152 x 1024 layer in 64 x 64 tiles -- pushed the plan towards 128 x 128 and made every shape
        //  but two slower, the weight gradients by 2-3 x: config 2 6.67 -> 7.58 / 8.88 ms in bf16; gpurun_out/r05k.  Dropped.)
        cost = flops / peak_flops_per_us / fill + partial_us + 2.0;
      }
      if (cost < best_cost * 0.97) {
        best_cost = cost;
        best = Plan{wi, wj, (int)real_split, (int)cps};
      }
    }
  }
  return best;
}

// scratch a GEMM of this shape may use for K slices (0 when it never splits)
static size_t plan_workspace(int I, int J, long long K, int max_split, bool always_reduce) {
  size_t worst = 0;
  for (int prec = 0; prec < 2; ++prec)
    for (int in_launch = 0; in_launch < (always_reduce ? 1 : 2); ++in_launch) {  // (either form of the slice sum may be the one that runs)
      const Plan p = plan_gemm(I, J, K, prec, max_split, ~(size_t)0, false, in_launch != 0);
      const size_t bytes = (p.nsplit > 1 || always_reduce) ? (size_t)p.nsplit * I * J * sizeof(float) : 0;
      worst = bytes > worst ? bytes : worst;
    }
  return worst;
}

// (ticket counters of the in-launch slice sums: cl3d::ticket_piece, cl3d_common.h)

constexpr int kMaxSplitFwd = 16;    // forward / input-gradient products: K = channels
constexpr int kMaxSplitWgrad = 512; // weight gradients: K = every point of the batch

// REDUCE_MODE 0: result through the output map; 1: PointWiseMLP weight-gradient merge (always via the reduce kernel)
// launch_plan: the product under plan p (+ the launch that sums its K slices, where the plan has one).  `a` by value: the
// measured-plan path below launches the same product under several plans.
template <int REDUCE_MODE>
static int launch_plan(GemmArgs a, const Plan &p, int precision, void *ws, const float *dwr, int Co, int C, hipStream_t st,
                       const char *who) {
  const int I = a.A.R, J = a.B.R;
  const bool always_reduce = REDUCE_MODE == 1;
  a.tiles_i = ceil_div(I, 64 * p.wi);
  a.tiles_j = ceil_div(J, 64 * p.wj);
  a.nsplit = p.nsplit;
  a.chunks_per_split = p.cps;
  a.partial = static_cast<float *>(ws);
  const bool reduce = p.nsplit > 1 || always_reduce;
  OutMap final_out = a.out;
  if (reduce && p.nsplit == 1) {  // one slice, but the result still goes through the reduce kernel's mapping
    a.out = OutMap{};
    a.out.D = a.partial; a.out.si = J; a.out.sj = 1;
  }
  const long long blocks = (long long)a.tiles_i * a.tiles_j * p.nsplit;
  if (blocks > 0x7fffffffLL) return fail(CL3D_E_UNSUPPORTED, "%s: grid too large", who);
  a.tickets = nullptr;
  a.vec_out = 0;
  if (REDUCE_MODE == 0 && in_launch_sum_applies(p.wi, p.wj, p.nsplit, fused_slice_sum())) {
    a.tickets = ticket_piece((size_t)a.tiles_i * a.tiles_j, st);
    a.vec_out = J % 4 == 0 && final_out.sj == 1 && final_out.si % 4 == 0 &&
                (final_out.fold_n == 0 || (final_out.fold_n % 4 == 0 && final_out.sb % 4 == 0)) &&
                (reinterpret_cast<uintptr_t>(final_out.D) & 15u) == 0 && (reinterpret_cast<uintptr_t>(a.partial) & 15u) == 0 &&
                (!final_out.ep_res || (reinterpret_cast<uintptr_t>(final_out.ep_res) & 15u) == 0);
  }
  if (precision == PREC_BF16) {
    launch_modes<PREC_BF16>(a, p.wi, p.wj, (int)blocks, st);
  } else {
    launch_modes<PREC_F32>(a, p.wi, p.wj, (int)blocks, st);
  }
  int rc = check_launch(who);
  if (rc != CL3D_OK || !reduce || a.tickets != nullptr) return rc;
  if (REDUCE_MODE == 0 && p.nsplit <= 16 && J % 4 == 0 && final_out.sj == 1 && final_out.si % 4 == 0 &&
      (final_out.fold_n == 0 || (final_out.fold_n % 4 == 0 && final_out.sb % 4 == 0)) &&
      (reinterpret_cast<uintptr_t>(final_out.D) & 15u) == 0 &&
      (!final_out.ep_res || (reinterpret_cast<uintptr_t>(final_out.ep_res) & 15u) == 0)) {
    long long g4 = ((long long)I * J / 4 + 255) / 256;
    if (g4 > 8192) g4 = 8192;
    hipLaunchKernelGGL(gemm_reduce4_kernel, dim3((unsigned)g4), dim3(256), 0, st, a.partial, p.nsplit, I, J, final_out);
    return check_launch(who);
  }
  const long long n_el = REDUCE_MODE == 0 ? (long long)I * J : (long long)Co * C;
  const bool vec4 = (REDUCE_MODE == 0 ? J % 4 == 0 : C % 4 == 0) && ((long long)I * J) % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(a.partial) & 15u) == 0;
  if (REDUCE_MODE == 1 && vec4 && p.nsplit <= 16) {
    long long g4 = (n_el / 4 + 255) / 256;
    if (g4 > 8192) g4 = 8192;
    hipLaunchKernelGGL(gemm_reduce_w4_kernel, dim3((unsigned)g4), dim3(256), 0, st, a.partial, p.nsplit, (long long)I * J,
                       final_out.D, dwr, Co, C);
    return check_launch(who);
  }
  long long grid = (n_el + (vec4 ? 63 : 15)) / (vec4 ? 64 : 16);
  if (grid > 16384) grid = 16384;
  if (vec4)
    hipLaunchKernelGGL((gemm_reduce_kernel<REDUCE_MODE, 4>), dim3((unsigned)grid), dim3(256), 0, st, a.partial, p.nsplit, I,
                       J, final_out, dwr, Co, C);
  else
    hipLaunchKernelGGL((gemm_reduce_kernel<REDUCE_MODE, 1>), dim3((unsigned)grid), dim3(256), 0, st, a.partial, p.nsplit, I,
                       J, final_out, dwr, Co, C);
  return check_launch(who);
}

// ---- plans by measurement (round 6, opt-in: cl3d_gemm_autotune(1)).  The launch-time model above is a fit with 15 % rms;
// over the config-2 backbone's 72 products the plans it picks cost 8-9 % more than the best plan of each
// (profiles/r06/gemm_plan_sweep_*_xcd.jsonl).  With the switch on, the FIRST call of a product outside stream capture --
// key: extents, operand modes, precision, prologue / epilogue, slice budget, scratch size -- runs every plan the model
// prices within 2.5x of its best a few times between two events on the caller's stream (the product is a pure function
// of its operands: running it again changes nothing), and the winner is kept for the process.  Inside a capture a product
// never seen before takes the model's plan.  Off by default: a different plan is a different summation order of the K
// slices, so with the switch on two PROCESSES may differ in the last bits of a result (one process never does: a key
// keeps its plan); bench.py / scripts/bench_backbone.py turn it on and say so in their lines.
struct TuneKey {
  long long K;
  size_t ws_bytes;
  int I, J, precision, am, bm, mode, max_split, flags;
  bool operator<(const TuneKey &o) const {
    return std::tie(K, ws_bytes, I, J, precision, am, bm, mode, max_split, flags) <
           std::tie(o.K, o.ws_bytes, o.I, o.J, o.precision, o.am, o.bm, o.mode, o.max_split, o.flags);
  }
};
static std::atomic<int> g_autotune{0};
static std::mutex g_tune_mu;
static std::map<TuneKey, Plan> g_tuned;
static std::atomic<long long> g_tune_counts[2];  // products measured, products whose measured plan differs from the model's

static std::vector<Plan> candidate_plans(int I, int J, long long K, int precision, int max_split, size_t ws_bytes,
                                         bool no_big_tile, bool in_launch_sum, const Plan &model) {
  const int cand[4][2] = {{2, 2}, {2, 1}, {1, 2}, {1, 1}};
  const double budget = 2.5 * bf16_launch_us(I, J, K, model.wi, model.wj, model.cps, model.nsplit, in_launch_sum);
  std::vector<Plan> out;
  for (int c = 0; c < 4; ++c) {
    const int wi = cand[c][0], wj = cand[c][1];
    if (no_big_tile && wi * wj == 4) continue;
    const int kc = gemm_kc(precision, wi, wj);
    const long long chunks = (K + kc - 1) / kc;
    const long long min_cps = precision == PREC_BF16 ? 2 : 4;
    for (long long split = 1; split <= max_split && split <= (chunks >= min_cps ? chunks / min_cps : 1); split += (split < 4 ? 1 : split / 2)) {
      if (split > 1 && (size_t)split * I * J * sizeof(float) > ws_bytes) break;
      const long long cps = (chunks + split - 1) / split;
      const long long real_split = (chunks + cps - 1) / cps;
      // (the bf16 fit prices both precisions here: only a coarse screen -- a weight gradient in one slice is 50x off)
      if (bf16_launch_us(I, J, K, wi, wj, cps, real_split, in_launch_sum) > budget) continue;
      bool seen = false;
      for (const Plan &q : out) seen = seen || (q.wi == wi && q.wj == wj && q.nsplit == (int)real_split);
      if (!seen) out.push_back(Plan{wi, wj, (int)real_split, (int)cps});
    }
  }
  return out;
}

template <int REDUCE_MODE>
static Plan measured_plan(const GemmArgs &a, const Plan &model, int precision, int max_split, void *ws, size_t ws_bytes,
                          bool no_big_tile, bool in_launch, const float *dwr, int Co, int C, hipStream_t st, const char *who) {
  constexpr int kReps = 3;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    (void)hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    return model;
  }
  Plan best = model;
  float best_ms = 1e30f, model_ms = 1e30f;
  std::vector<Plan> cands = candidate_plans(a.A.R, a.B.R, a.K, precision, max_split, ws_bytes, no_big_tile, in_launch, model);
  bool has_model = false;
  for (const Plan &q : cands) has_model = has_model || (q.wi == model.wi && q.wj == model.wj && q.nsplit == model.nsplit);
  if (!has_model) cands.push_back(model);
  for (const Plan &q : cands) {
    bool ok = launch_plan<REDUCE_MODE>(a, q, precision, ws, dwr, Co, C, st, who) == CL3D_OK;  // (code object, caches)
    ok = ok && hipEventRecord(e0, st) == hipSuccess;
    for (int r = 0; ok && r < kReps; ++r) ok = launch_plan<REDUCE_MODE>(a, q, precision, ws, dwr, Co, C, st, who) == CL3D_OK;
    float ms = 0.f;
    ok = ok && hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
         hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      best = model;
      break;
    }
    if (q.wi == model.wi && q.wj == model.wj && q.nsplit == model.nsplit) model_ms = ms;
    if (ms < best_ms) {
      best_ms = ms;
      best = q;
    }
  }
  if (best_ms > 0.98f * model_ms) best = model;  // (within the timing's noise of the model's plan: keep that)
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return best;
}

template <int REDUCE_MODE>
static int run_gemm(GemmArgs &a, int precision, int max_split, void *ws, size_t ws_bytes, const float *dwr, int Co, int C,
                    hipStream_t st, const char *who) {
  const int I = a.A.R, J = a.B.R;
  if (I == 0 || J == 0) return CL3D_OK;
  const bool always_reduce = REDUCE_MODE == 1;
  if (always_reduce && (!ws || ws_bytes < (size_t)I * J * sizeof(float)))
    return fail(CL3D_E_WORKSPACE, "%s: workspace %zu < %zu", who, ws_bytes, plan_workspace(I, J, a.K, max_split, true));
  const int am = stage_mode(a.A), bm = stage_mode(a.B);
  const bool vec_pair = (am == STAGE_VEC_RC && bm == STAGE_VEC_KC) || (am == STAGE_VEC_KC && bm == STAGE_VEC_RC) ||
                        (am == STAGE_VEC_RC && bm == STAGE_VEC_RC) || (am == STAGE_VEC_KC && bm == STAGE_VEC_KC);
  const bool kc_pair_f32 = precision != PREC_BF16 && am == STAGE_VEC_KC && bm == STAGE_VEC_KC;
  const bool in_launch = REDUCE_MODE == 0 && fused_slice_sum();
  const bool no_big_tile = !vec_pair || kc_pair_f32;
  Plan p = plan_gemm(I, J, a.K, precision, ws ? max_split : 1, ws ? ws_bytes : 0, no_big_tile, in_launch);
  if (g_autotune.load(std::memory_order_relaxed) != 0) {
    const int flags = (a.A.pro_scale ? 1 : 0) | (a.B.pro_scale ? 2 : 0) | (a.out.ep_scale ? 4 : 0) | (a.out.ep_res ? 8 : 0) |
                      (a.out.ep_relu ? 16 : 0) | (a.out.fold_n ? 32 : 0) | (a.A.fold ? 64 : 0) | (a.B.fold ? 128 : 0);
    const TuneKey key{a.K, ws ? ws_bytes : 0, I, J, precision, am, bm, REDUCE_MODE, ws ? max_split : 1, flags};
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    std::lock_guard<std::mutex> lock(g_tune_mu);
    auto it = g_tuned.find(key);
    if (it != g_tuned.end()) {
      p = it->second;
    } else if (!capturing) {
      const Plan m = measured_plan<REDUCE_MODE>(a, p, precision, ws ? max_split : 1, ws, ws ? ws_bytes : 0, no_big_tile,
                                                in_launch, dwr, Co, C, st, who);
      g_tune_counts[0] += 1;
      if (m.wi != p.wi || m.wj != p.wj || m.nsplit != p.nsplit) g_tune_counts[1] += 1;
      g_tuned[key] = m;
      p = m;
    }
  }
#ifdef CL3D_GEMM_PLAN_ENV  // variant builds of scripts/micro/gemm_plan_sweep.py only: CL3D_GEMM_FORCE="wi,wj,split"
  if (const char *force = getenv("CL3D_GEMM_FORCE")) {
    int fwi = 0, fwj = 0, fsplit = 0;
    if (sscanf(force, "%d,%d,%d", &fwi, &fwj, &fsplit) == 3 && (fwi == 1 || fwi == 2) && (fwj == 1 || fwj == 2) && fsplit >= 1) {
      const int kc = gemm_kc(precision, fwi, fwj);
      const long long chunks = (a.K + kc - 1) / kc;
      if (fsplit > chunks) fsplit = (int)chunks;
      while (fsplit > 1 && (!ws || (size_t)fsplit * I * J * sizeof(float) > ws_bytes)) --fsplit;
      const long long cps = (chunks + fsplit - 1) / fsplit;
      p = Plan{fwi, fwj, (int)((chunks + cps - 1) / cps), (int)cps};
    }
  }
#endif
  return launch_plan<REDUCE_MODE>(a, p, precision, ws, dwr, Co, C, st, who);
}

static int round_up_grid(int n) {
  const int g = ceil_div(n, 256);
  return g < 1 ? 1 : (g > 2048 ? 2048 : g);
}

// scratch of the three products of a per-point contraction  out[Co'] <- in[C']  over nb clouds of n points
size_t gemm_family_workspace(int nb, int n, int rows_out, int rows_in, bool merge) {
  const long long P = (long long)nb * n;
  if (P > 0x7fffffffLL) return 0;
  size_t w = plan_workspace(rows_out, rows_in, P, kMaxSplitWgrad, merge);                    // weight gradient
  const size_t f = plan_workspace((int)P, rows_out, rows_in, kMaxSplitFwd, false);           // forward (either orientation:
  const size_t f2 = plan_workspace(rows_out, (int)P, rows_in, kMaxSplitFwd, false);          //  points on i or on j)
  const size_t d = plan_workspace(rows_in, (int)P, rows_out, kMaxSplitFwd, false);           // input gradient
  w = f > w ? f : w;
  w = f2 > w ? f2 : w;
  w = d > w ? d : w;
  if (merge) {  // pwmlp_point_grads_kernel: one partial d wcat tile per workgroup
    const size_t g = (size_t)kPointGradsGrid * rows_out * rows_in * sizeof(float);
    w = g > w ? g : w;
  }
  return w;
}

// the workgroups' partial d wcat tiles [nparts][2Co][C] -> d W [Co, 3+2C] (d W_c = bot, d W_d = top - bot, d W_r copied).
// A workgroup owns four 16-byte pieces of the top half and the same four of the bottom half; its 64 thread groups take
// the partials g, g+64, g+128, ... -- for the 256 partials of the metric shape that is 8 independent 16-byte loads per
// thread, all in flight at once (gemm_reduce_kernel<1> walks its slices 16 apart in a loop of dependent round trips:
// 8.6-12 us for these 8 MB) -- and the 64 sub-sums are added in group order through LDS: the order depends on nparts only.
__global__ __launch_bounds__(256) void pwmlp_dw_reduce_kernel(const float *__restrict__ part, int nparts, int Co, int C,
                                                              const float *__restrict__ dwr, float *__restrict__ dW) {
  __shared__ float4 s_sum[2][64][4];
  const int q = threadIdx.x & 3, g = threadIdx.x >> 2;
  const long long half = (long long)Co * C, IJ = 2 * half;
  const long long e = ((long long)blockIdx.x * 4 + q) * 4;  // first of this thread's four elements (o, c .. c+3)
  float4 top = make_float4(0.f, 0.f, 0.f, 0.f), bot = top;
  if (e < half) {
#pragma unroll 4
    for (int p = g; p < nparts; p += 64) {
      const float4 t = *reinterpret_cast<const float4 *>(part + (size_t)p * IJ + e);
      const float4 u = *reinterpret_cast<const float4 *>(part + (size_t)p * IJ + half + e);
      top.x += t.x; top.y += t.y; top.z += t.z; top.w += t.w;
      bot.x += u.x; bot.y += u.y; bot.z += u.z; bot.w += u.w;
    }
  }
  s_sum[0][g][q] = top;
  s_sum[1][g][q] = bot;
  __syncthreads();
  if (threadIdx.x < 32) {  // 2 halves x 4 pieces x 4 components: one lane per float, 64 adds in group order
    const int h = threadIdx.x >> 4, qq = (threadIdx.x >> 2) & 3, comp = threadIdx.x & 3;
    const float *src = reinterpret_cast<const float *>(&s_sum[h][0][qq]) + comp;
    float acc = src[0];
#pragma unroll 8
    for (int k = 1; k < 64; ++k) acc += src[k * 16];
    reinterpret_cast<float *>(&s_sum[h][0][qq])[comp] = acc;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int qq = threadIdx.x >> 2, comp = threadIdx.x & 3;
    const long long ev = ((long long)blockIdx.x * 4 + qq) * 4 + comp;
    if (ev < half) {
      const float t = reinterpret_cast<const float *>(&s_sum[0][0][qq])[comp];
      const float bt = reinterpret_cast<const float *>(&s_sum[1][0][qq])[comp];
      const int oo = (int)(ev / C), c = (int)(ev - (long long)oo * C);
      const int ld = 3 + 2 * C;
      dW[(size_t)oo * ld + 3 + c] = bt;          // d W_c
      dW[(size_t)oo * ld + 3 + C + c] = t - bt;  // d W_d
      if (c < 3) dW[(size_t)oo * ld + c] = dwr ? dwr[oo * 3 + c] : 0.f;
    }
  }
}

// ---- host side of pwmlp_point_grads_kernel
static bool point_grads_covers(int B, int C, int N, int Co, int precision) {
  return precision == PREC_F32 && B >= 1 && C >= 4 && C <= 64 && C % 4 == 0 && Co >= 2 && 2 * Co <= 128 && Co % 2 == 0 &&
         N % kPgTP == 0 && N % 4 == 0;
}

static int launch_point_grads(const float *features, const float *pro_scale, const float *pro_shift, const float *dght,
                              const float *wcat, const float *dwr, int B, int C, int N, int Co, float *dfeat, float *dW,
                              void *ws, size_t ws_bytes, hipStream_t st, const char *who) {
  const int J = 2 * Co;
  const long long tiles = (long long)B * (N / kPgTP);
  const int grid = (int)(tiles < kPointGradsGrid ? tiles : kPointGradsGrid);
  const size_t need = dW != nullptr ? (size_t)grid * J * C * sizeof(float) : 0;
  if (dW != nullptr && (ws == nullptr || ws_bytes < need))
    return fail(CL3D_E_WORKSPACE, "%s: workspace %zu < %zu", who, ws_bytes, need);
  if (!aligned16(dght) || !aligned16(features) || !aligned16(ws) || (pro_scale != nullptr && !aligned16(pro_scale)))
    return fail(CL3D_E_INVALID, "%s: operands must be 16-byte aligned", who);
  const size_t lds = (size_t)2 * kPgBufFloats * sizeof(float);
  PointGradArgs a{};
  a.dght = dght; a.F = features; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.wcat = wcat;
  a.dfeat = dfeat; a.partial = dW != nullptr ? static_cast<float *>(ws) : nullptr;
  a.C = C; a.J = J; a.N = N; a.tiles = (int)tiles; a.tiles_per_cloud = N / kPgTP;
  const bool full = C == 64 && J == 128, pro = pro_scale != nullptr;
  int rc = CL3D_OK;
#define CL3D_PG(FULL_, PRO_)                                                                                          \
  do {                                                                                                                \
    static std::atomic<unsigned long long> granted{0};                                                                \
    rc = lds_opt_in(granted, reinterpret_cast<const void *>(pwmlp_point_grads_kernel<FULL_, PRO_>), lds, who);         \
    if (rc == CL3D_OK) hipLaunchKernelGGL((pwmlp_point_grads_kernel<FULL_, PRO_>), dim3(grid), dim3(256), lds, st, a); \
  } while (0)
  if (full && pro) CL3D_PG(true, true);
  else if (full) CL3D_PG(true, false);
  else if (pro) CL3D_PG(false, true);
  else CL3D_PG(false, false);
#undef CL3D_PG
  if (rc != CL3D_OK) return rc;
  rc = check_launch(who);
  if (rc != CL3D_OK || dW == nullptr) return rc;
  const int rgrid = ceil_div(Co * C, 16);  // (C % 4 == 0: a 16-byte piece never straddles a row of d wcat)
  hipLaunchKernelGGL(pwmlp_dw_reduce_kernel, dim3(rgrid), dim3(256), 0, st, a.partial, grid, Co, C, dwr, dW);
  return check_launch(who);
}

}  // namespace cl3d

using namespace cl3d;

#define GEMM_COMMON_CHECKS(who)                                                                       \
  CL3D_REQUIRE(B >= 0 && C >= 1 && N >= 1 && Co >= 1, who ": bad sizes");                            \
  CL3D_REQUIRE(precision == 0 || precision == 1, who ": precision must be 0 (f32) or 1 (bf16)");      \
  if ((long long)B * N * (long long)(2 * Co > C ? 2 * Co : C) > 0x7fffffffLL)                         \
    return fail(CL3D_E_UNSUPPORTED, who ": tensor too large (32-bit element offsets)");

static int point_gemm_fwd(const float *features, const float *pro_scale, const float *pro_shift, const float *W, int B,
                          int C, int N, int Co, int precision, float *ght, float *wr, float *wcat, void *ws,
                          size_t ws_bytes, hipStream_t st, const char *who) {
  hipLaunchKernelGGL(pwmlp_weights_kernel, dim3(round_up_grid(Co * (3 + 2 * C))), dim3(256), 0, st, W, Co, C, wr, wcat);
  const int rc = check_launch(who);
  if (rc != CL3D_OK || B == 0) return rc;
  if (precision == PREC_F32 && launch_rows_nolds(features, pro_scale, pro_shift, wcat, ght, B, C, N, Co, st)) return check_launch(who);
  if (precision == PREC_BF16 && launch_rows_nolds_bf16(features, pro_scale, pro_shift, wcat, ght, B, C, N, Co, st)) return check_launch(who);
  GemmArgs a{};  // D[i = (cloud, point)][j = o] = sum_c F[c][point] wcat[o][c]  ->  ght [B*N, 2Co]
  a.A = channel_major(features, B, C, N, true);
  set_prologue(a.A, pro_scale, pro_shift, 0);
  a.B = plain(wcat, C, 1, 2 * Co, C);
  a.out.D = ght; a.out.si = 2 * Co; a.out.sj = 1;
  a.K = C;
  return run_gemm<0>(a, precision, kMaxSplitFwd, ws, ws_bytes, nullptr, 0, 0, st, who);
}

extern "C" int cl3d_pwmlp_point_gemm_fwd(const float *features, const float *W, int B, int C, int N, int Co,
                                         int precision, float *ght, float *wr, float *wcat, void *ws, size_t ws_bytes,
                                         cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("pwmlp_point_gemm_fwd");
  CL3D_REQUIRE(W && wcat && (B == 0 || (features && ght)), "pwmlp_point_gemm_fwd: null pointer");
  return point_gemm_fwd(features, nullptr, nullptr, W, B, C, N, Co, precision, ght, wr, wcat, ws, ws_bytes,
                        (hipStream_t)stream, "cl3d_pwmlp_point_gemm_fwd");
}

// the same product on features = max(scale[c] * x + shift[c], 0): the BatchNorm + ReLU of the bottleneck's conv1
// (backbones/resnet.py:32-34) applied while the tile is staged, so the activated tensor is never written
extern "C" int cl3d_pwmlp_point_gemm_fwd_pro(const float *x, const float *scale, const float *shift, const float *W,
                                             int B, int C, int N, int Co, int precision, float *ght, float *wr,
                                             float *wcat, void *ws, size_t ws_bytes, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("pwmlp_point_gemm_fwd_pro");
  CL3D_REQUIRE(W && wcat && scale && shift && (B == 0 || (x && ght)), "pwmlp_point_gemm_fwd_pro: null pointer");
  return point_gemm_fwd(x, scale, shift, W, B, C, N, Co, precision, ght, wr, wcat, ws, ws_bytes, (hipStream_t)stream,
                        "cl3d_pwmlp_point_gemm_fwd_pro");
}

extern "C" int cl3d_pwmlp_point_gemm_bwd_data(const float *dght, const float *wcat, int B, int C, int N, int Co,
                                              int precision, float *dfeatures, void *ws, size_t ws_bytes,
                                              cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("pwmlp_point_gemm_bwd_data");
  CL3D_REQUIRE(wcat && (B == 0 || (dght && dfeatures)), "pwmlp_point_gemm_bwd_data: null pointer");
  if (B == 0) return CL3D_OK;
  GemmArgs a{};  // D[i = c][j = (cloud, point)] = sum_o wcat[o][c] dght[point][o]  ->  d features [B, C, N]
  a.A = plain(wcat, 1, C, C, 2 * Co);
  a.B = plain(dght, 2 * Co, 1, B * N, 2 * Co);
  a.out.D = dfeatures; a.out.si = N; a.out.sj = 1; a.out.fold_n = B > 1 ? N : 0; a.out.fold_shift = log2_exact(N); a.out.sb = (long long)C * N;
  a.K = 2 * Co;
  return run_gemm<0>(a, precision, kMaxSplitFwd, ws, ws_bytes, nullptr, 0, 0, (hipStream_t)stream,
                     "cl3d_pwmlp_point_gemm_bwd_data");
}

static int point_gemm_bwd_weight(const float *features, const float *pro_scale, const float *pro_shift,
                                 const float *dght, const float *dwr, int B, int C, int N, int Co, int precision,
                                 float *dW, void *ws, size_t ws_bytes, hipStream_t st, const char *who) {
  GemmArgs a{};  // D[i = o][j = c] = sum_(cloud, point) dght[point][o] F[c][point]  ->  d wcat [2Co, C] per slice
  a.A = plain(dght, 1, 2 * Co, 2 * Co, B * N);
  a.B = channel_major(features, B, C, N, false);
  set_prologue(a.B, pro_scale, pro_shift, 1);
  a.K = B * N;
  a.out.D = dW;
  return run_gemm<1>(a, precision, kMaxSplitWgrad, ws, ws_bytes, dwr, Co, C, st, who);
}

extern "C" int cl3d_pwmlp_point_gemm_bwd_weight(const float *features, const float *dght, const float *dwr, int B,
                                                int C, int N, int Co, int precision, float *dW, void *ws,
                                                size_t ws_bytes, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("pwmlp_point_gemm_bwd_weight");
  CL3D_REQUIRE(B >= 1 && features && dght && dW, "pwmlp_point_gemm_bwd_weight: null pointer");
  return point_gemm_bwd_weight(features, nullptr, nullptr, dght, dwr, B, C, N, Co, precision, dW, ws, ws_bytes,
                               (hipStream_t)stream, "cl3d_pwmlp_point_gemm_bwd_weight");
}

extern "C" int cl3d_pwmlp_point_gemm_bwd_weight_pro(const float *x, const float *scale, const float *shift,
                                                    const float *dght, const float *dwr, int B, int C, int N, int Co,
                                                    int precision, float *dW, void *ws, size_t ws_bytes,
                                                    cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("pwmlp_point_gemm_bwd_weight_pro");
  CL3D_REQUIRE(B >= 1 && x && scale && shift && dght && dW, "pwmlp_point_gemm_bwd_weight_pro: null pointer");
  return point_gemm_bwd_weight(x, scale, shift, dght, dwr, B, C, N, Co, precision, dW, ws, ws_bytes, (hipStream_t)stream,
                               "cl3d_pwmlp_point_gemm_bwd_weight_pro");
}

extern "C" int cl3d_pwmlp_point_gemm_bwd_fused(int B, int C, int N, int Co, int precision) {
  return point_grads_covers(B, C, N, Co, precision) && (long long)B * (N / kPgTP) <= 0x7fffffffLL ? 1 : 0;
}

// both gradients of the per-point product: ONE kernel over d ght where pwmlp_point_grads_kernel covers the shape
// (cl3d_pwmlp_point_gemm_bwd_fused), the two products one after the other on `stream` otherwise
extern "C" int cl3d_pwmlp_point_gemm_bwd(const float *features, const float *scale, const float *shift,
                                         const float *dght, const float *wcat, const float *dwr, int B, int C, int N,
                                         int Co, int precision, float *dfeatures, float *dW, void *ws, size_t ws_bytes,
                                         cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("pwmlp_point_gemm_bwd");
  CL3D_REQUIRE((scale == nullptr) == (shift == nullptr), "pwmlp_point_gemm_bwd: scale and shift go together");
  CL3D_REQUIRE(dfeatures != nullptr || dW != nullptr, "pwmlp_point_gemm_bwd: no output asked for");
  CL3D_REQUIRE(B == 0 || (dght && wcat && (dW == nullptr || features)), "pwmlp_point_gemm_bwd: null pointer");
  if (B == 0) return dW == nullptr ? CL3D_OK : fail(CL3D_E_INVALID, "pwmlp_point_gemm_bwd: a weight gradient needs B >= 1");
  if (cl3d_pwmlp_point_gemm_bwd_fused(B, C, N, Co, precision) && features != nullptr)
    return launch_point_grads(features, scale, shift, dght, wcat, dwr, B, C, N, Co, dfeatures, dW, ws, ws_bytes,
                              (hipStream_t)stream, "cl3d_pwmlp_point_gemm_bwd");
  if (dfeatures != nullptr) {
    const int rc = cl3d_pwmlp_point_gemm_bwd_data(dght, wcat, B, C, N, Co, precision, dfeatures, ws, ws_bytes, stream);
    if (rc != CL3D_OK) return rc;
  }
  if (dW != nullptr)
    return point_gemm_bwd_weight(features, scale, shift, dght, dwr, B, C, N, Co, precision, dW, ws, ws_bytes,
                                 (hipStream_t)stream, "cl3d_pwmlp_point_gemm_bwd");
  return CL3D_OK;
}

// ---- the 1x1 Conv1d layers around the operator (backbones/resnet.py:32-39,58-66), channel-major in and out ---------
static int conv_forward(const float *x, const float *W, const float *scale, const float *shift, const float *residual,
                        int relu, int B, int C, int N, int Co, int precision, float *y, void *ws, size_t ws_bytes,
                        hipStream_t st, const char *who) {
  GemmArgs a{};  // D[i = o][j = (cloud, point)] = sum_c W[o][c] x[c][point]
  a.A = plain(W, C, 1, Co, C);
  a.B = channel_major(x, B, C, N, true);
  a.out.D = y; a.out.si = N; a.out.sj = 1; a.out.fold_n = B > 1 ? N : 0; a.out.fold_shift = log2_exact(N); a.out.sb = (long long)Co * N;
  a.out.ep_scale = scale; a.out.ep_shift = shift; a.out.ep_res = residual; a.out.ep_relu = relu;
  a.K = C;
  return run_gemm<0>(a, precision, kMaxSplitFwd, ws, ws_bytes, nullptr, 0, 0, st, who);
}

extern "C" int cl3d_conv1x1_fwd(const float *x, const float *W, int B, int C, int N, int Co, int precision, float *y,
                                void *ws, size_t ws_bytes, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("conv1x1_fwd");
  CL3D_REQUIRE(W && (B == 0 || (x && y)), "conv1x1_fwd: null pointer");
  if (B == 0) return CL3D_OK;
  return conv_forward(x, W, nullptr, nullptr, nullptr, 0, B, C, N, Co, precision, y, ws, ws_bytes, (hipStream_t)stream,
                      "cl3d_conv1x1_fwd");
}

// inference: y = act(scale[o] * (W x)[o] + shift[o] + residual) in the GEMM's epilogue -- the BatchNorm of eval mode
// folded into a per-channel affine map (scale = gamma / sqrt(var + eps), shift = beta - mean * scale), the shortcut
// add and the ReLU of backbones/resnet.py:58-66 without a pass of their own
extern "C" int cl3d_conv1x1_bn_act_fwd(const float *x, const float *W, const float *scale, const float *shift,
                                       const float *residual, int relu, int B, int C, int N, int Co, int precision,
                                       float *y, void *ws, size_t ws_bytes, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("conv1x1_bn_act_fwd");
  CL3D_REQUIRE(W && (B == 0 || (x && y)) && (!scale == !shift), "conv1x1_bn_act_fwd: null pointer");
  if (B == 0) return CL3D_OK;
  return conv_forward(x, W, scale, shift, residual, relu, B, C, N, Co, precision, y, ws, ws_bytes, (hipStream_t)stream,
                      "cl3d_conv1x1_bn_act_fwd");
}

extern "C" int cl3d_conv1x1_bwd_data(const float *dy, const float *W, int B, int C, int N, int Co, int precision,
                                     float *dx, void *ws, size_t ws_bytes, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("conv1x1_bwd_data");
  CL3D_REQUIRE(W && (B == 0 || (dy && dx)), "conv1x1_bwd_data: null pointer");
  if (B == 0) return CL3D_OK;
  GemmArgs a{};  // D[i = c][j = (cloud, point)] = sum_o W[o][c] dy[o][point]
  a.A = plain(W, 1, C, C, Co);
  a.B = channel_major(dy, B, Co, N, true);
  a.out.D = dx; a.out.si = N; a.out.sj = 1; a.out.fold_n = B > 1 ? N : 0; a.out.fold_shift = log2_exact(N); a.out.sb = (long long)C * N;
  a.K = Co;
  return run_gemm<0>(a, precision, kMaxSplitFwd, ws, ws_bytes, nullptr, 0, 0, (hipStream_t)stream, "cl3d_conv1x1_bwd_data");
}

extern "C" int cl3d_conv1x1_bwd_weight(const float *x, const float *dy, int B, int C, int N, int Co, int precision,
                                       float *dW, void *ws, size_t ws_bytes, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("conv1x1_bwd_weight");
  CL3D_REQUIRE(B >= 1 && x && dy && dW, "conv1x1_bwd_weight: null pointer");
  GemmArgs a{};  // D[i = o][j = c] = sum_(cloud, point) dy[o][point] x[c][point]
  a.A = channel_major(dy, B, Co, N, false);
  a.B = channel_major(x, B, C, N, false);
  a.K = B * N;
  a.out.D = dW; a.out.si = C; a.out.sj = 1;
  return run_gemm<0>(a, precision, kMaxSplitWgrad, ws, ws_bytes, nullptr, 0, 0, (hipStream_t)stream,
                     "cl3d_conv1x1_bwd_weight");
}

// ---- the convolution that FOLLOWS the operator (backbones/resnet.py:58: conv2), fed by the operator's point-major
// rows [B, N, C] (its per-(query, channel) pre-activations) with the operator's BatchNorm + ReLU as the operand
// prologue: the channel-major activated tensor between the operator and conv2 is never written.  The input gradient
// comes back as rows too -- the gradient with respect to the ACTIVATED input, which the operator's backward expects.
extern "C" int cl3d_conv1x1_rows_fwd(const float *x_rows, const float *scale, const float *shift, const float *W, int B,
                                     int C, int N, int Co, int precision, float *y, void *ws, size_t ws_bytes,
                                     cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("conv1x1_rows_fwd");
  CL3D_REQUIRE(W && scale && shift && (B == 0 || (x_rows && y)), "conv1x1_rows_fwd: null pointer");
  if (B == 0) return CL3D_OK;
  GemmArgs a{};  // D[i = o][j = (cloud, point)] = sum_c W[o][c] act(x[(cloud, point)][c])
  a.A = plain(W, C, 1, Co, C);
  a.B = plain(x_rows, C, 1, B * N, C);
  set_prologue(a.B, scale, shift, 0);
  a.out.D = y; a.out.si = N; a.out.sj = 1; a.out.fold_n = B > 1 ? N : 0; a.out.fold_shift = log2_exact(N); a.out.sb = (long long)Co * N;
  a.K = C;
  return run_gemm<0>(a, precision, kMaxSplitFwd, ws, ws_bytes, nullptr, 0, 0, (hipStream_t)stream, "cl3d_conv1x1_rows_fwd");
}

extern "C" int cl3d_conv1x1_rows_bwd_data(const float *dy, const float *W, int B, int C, int N, int Co, int precision,
                                          float *dx_rows, void *ws, size_t ws_bytes, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("conv1x1_rows_bwd_data");
  CL3D_REQUIRE(W && (B == 0 || (dy && dx_rows)), "conv1x1_rows_bwd_data: null pointer");
  if (B == 0) return CL3D_OK;
  GemmArgs a{};  // D[i = (cloud, point)][j = c] = sum_o dy[o][point] W[o][c]  ->  rows [B*N, C]
  a.A = channel_major(dy, B, Co, N, true);
  a.B = plain(W, 1, C, C, Co);
  a.out.D = dx_rows; a.out.si = C; a.out.sj = 1;
  a.K = Co;
  return run_gemm<0>(a, precision, kMaxSplitFwd, ws, ws_bytes, nullptr, 0, 0, (hipStream_t)stream,
                     "cl3d_conv1x1_rows_bwd_data");
}

extern "C" int cl3d_conv1x1_rows_bwd_weight(const float *x_rows, const float *scale, const float *shift, const float *dy,
                                            int B, int C, int N, int Co, int precision, float *dW, void *ws,
                                            size_t ws_bytes, cl3d_stream_t stream) {
  GEMM_COMMON_CHECKS("conv1x1_rows_bwd_weight");
  CL3D_REQUIRE(B >= 1 && x_rows && scale && shift && dy && dW, "conv1x1_rows_bwd_weight: null pointer");
  GemmArgs a{};  // D[i = o][j = c] = sum_(cloud, point) dy[o][point] act(x[(cloud, point)][c])
  a.A = channel_major(dy, B, Co, N, false);
  a.B = plain(x_rows, 1, C, C, B * N);
  set_prologue(a.B, scale, shift, 1);
  a.K = B * N;
  a.out.D = dW; a.out.si = C; a.out.sj = 1;
  return run_gemm<0>(a, precision, kMaxSplitWgrad, ws, ws_bytes, nullptr, 0, 0, (hipStream_t)stream,
                     "cl3d_conv1x1_rows_bwd_weight");
}

// ---- plans by measurement: the switch (process-wide; see measured_plan above) and its counters
extern "C" int cl3d_gemm_autotune(int enable) { return cl3d::g_autotune.exchange(enable != 0 ? 1 : 0); }

extern "C" int cl3d_gemm_autotune_stats(long long *measured, long long *changed) {
  if (measured) *measured = cl3d::g_tune_counts[0].load();
  if (changed) *changed = cl3d::g_tune_counts[1].load();
  return CL3D_OK;
}

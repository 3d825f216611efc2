// bn_relu.hip -- BatchNorm1d + ReLU on channel-major [B,C,N] tensors: the output transform every
// LocalAggregation operator ends in (reference: local_aggregation_operators.py:40-45 `out_transform`, and the
// BN+ReLU halves of `out_conv`).  Streaming, HBM-bound (19 MB per tensor at the metric shape):
//   training forward   STATS (sum x, sum x^2 per (cloud, channel) row, double)  ->  FINALIZE (batch mean /
//                      variance, scale / shift, running-statistics update with nn.BatchNorm1d's rule)  ->
//                      APPLY  out = ReLU(scale * x + shift)
//   inference forward  APPLY with scale / shift from the running statistics
//   backward           BWD_STATS (dz = g gated by the ReLU, recomputed from x; sum dz, sum dz * xhat)  ->
//                      COEFFS (dx = A dz + Bc + D x: the BatchNorm backward is affine in x)  ->  BWD_APPLY
// Three launches each way instead of the library's BatchNorm + a separate ReLU (+ their backward kernels),
// and x is the only tensor kept for the backward pass (the ReLU mask and xhat are recomputed from it).
#include "cl3d_common.h"

namespace cl3d {

struct BnFinArgs {
  const double *partial;  // [G, C, 2]
  int G, C;
  double count;
  float eps, momentum;
  const float *gamma, *beta, *mean_in, *invstd_in;
  float *running_mean, *running_var;
  long long *num_batches_tracked;  // nn.BatchNorm's step counter, bumped by the block of channel 0 (may be null)
  float *o0, *o1, *o2, *o3, *o4;
};

struct BnArgs {
  const float *x, *g;          // [B,C,N]
  const float *scale, *shift, *mean, *invstd, *cA, *cB, *cD;
  float *out;                  // [B,C,N]
  double *partial;             // [B*chunks, C, 2]
  int B, C, N, chunks, span;   // a block reduces `span` points of one (cloud, channel) row
  unsigned *tickets;           // [C] or null.  Set: the workgroup that arrives last at a channel's ticket runs `fin`
  BnFinArgs fin;               //   for that channel (statistics kernels only) and no finalize launch follows
};

// The per-channel algebra behind a statistics pass, on the first 64 threads of a workgroup.
// MODE 0: batch statistics -> scale, shift, mean, invstd (+ running update);  MODE 1: backward coefficients.
// DEVICE_SCOPE: the partials were written by other workgroups of the SAME launch (round 6: the statistics kernels run
// this themselves, in the workgroup that arrives last at the channel's ticket -- no finalize launch); they are read with
// device-scope loads.  Same summation order either way (thread t adds blocks t, t + 64, ...; a butterfly over the wave).
template <int MODE, bool DEVICE_SCOPE>
__device__ __forceinline__ void bn_finalize_channel(const BnFinArgs &a, int c, int tid) {
  double s0 = 0.0, s1 = 0.0;
  for (int g = tid; g < a.G; g += 64) {
    const double *p = a.partial + ((size_t)g * a.C + c) * 2;
    if constexpr (DEVICE_SCOPE) {
      s0 += __hip_atomic_load(const_cast<double *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s1 += __hip_atomic_load(const_cast<double *>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      s0 += p[0];
      s1 += p[1];
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    s0 += __shfl_xor(s0, o, 64);
    s1 += __shfl_xor(s1, o, 64);
  }
  if (tid != 0) return;
  if (MODE == 0) {
    const double mean = s0 / a.count;
    double var = s1 / a.count - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double invstd = 1.0 / sqrt(var + (double)a.eps);
    const double scale = (double)a.gamma[c] * invstd;
    a.o0[c] = (float)scale;
    a.o1[c] = (float)((double)a.beta[c] - mean * scale);
    a.o2[c] = (float)mean;
    a.o3[c] = (float)invstd;
    if (a.running_mean != nullptr) {  // nn.BatchNorm1d: running = (1-m) running + m batch, unbiased variance
      const double unbiased = var * (a.count / (a.count > 1.0 ? a.count - 1.0 : 1.0));
      a.running_mean[c] = a.running_mean[c] * (1.0f - a.momentum) + a.momentum * (float)mean;
      a.running_var[c] = a.running_var[c] * (1.0f - a.momentum) + a.momentum * (float)unbiased;
    }
    if (c == 0 && a.num_batches_tracked != nullptr) *a.num_batches_tracked += 1;
  } else {  // dx = A dz + Bc + D x   (s0 = sum dz = d beta, s1 = sum dz * xhat = d gamma)
    const double invstd = (double)a.invstd_in[c], mean = (double)a.mean_in[c];
    const double A = (double)a.gamma[c] * invstd;
    const double D = -A * invstd * s1 / a.count;
    const double Bc = -A * s0 / a.count - D * mean;
    a.o0[c] = (float)A;
    a.o1[c] = (float)Bc;
    a.o2[c] = (float)D;
    a.o3[c] = (float)s1;
    a.o4[c] = (float)s0;
  }
}

// a workgroup's partial pair: plain stores (a finalize launch follows) or device-scope stores (another workgroup of this
// launch reads them)
__device__ __forceinline__ void bn_store_partial(double *p, double t0, double t1, bool device_scope) {
  if (device_scope) {
    __hip_atomic_store(p, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    p[0] = t0;
    p[1] = t1;
  }
}

// points of one (cloud, channel) row per workgroup of the statistics passes (2048 -- twice the workgroups, two 16-byte
// items per thread instead of four -- measured in round 3: no faster, 6.7 / 12.4 against 7.4 / 12.0 us)
constexpr int kBnSpan = 16384;

__device__ __forceinline__ double block_sum(double v, double *scratch) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[wave] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// MODE 0: forward statistics of x;  MODE 1: backward statistics (dz, dz*xhat)
template <int MODE>
__global__ __launch_bounds__(256) void bn_stats_kernel(BnArgs a) {
  __shared__ double scratch[4];
  const int c = blockIdx.x, part = blockIdx.y;
  const int b = part / a.chunks, n0 = (part - b * a.chunks) * a.span;
  const int n1 = n0 + a.span < a.N ? n0 + a.span : a.N;
  const float *xr = a.x + ((size_t)b * a.C + c) * a.N;
  const float *gr = MODE == 1 ? a.g + ((size_t)b * a.C + c) * a.N : nullptr;
  float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
  if (MODE == 1) {
    sc = a.scale[c]; sh = a.shift[c]; mu = a.mean[c]; is = a.invstd[c];
  }
  float s0 = 0.f, s1 = 0.f;  // <= 64 terms per thread between the double folds below
  double d0 = 0.0, d1 = 0.0;
  auto item = [&](float xv, float gv) {
    if (MODE == 0) {
      s0 += xv;
      s1 = __builtin_fmaf(xv, xv, s1);
    } else {
      const float dz = __builtin_fmaf(xv, sc, sh) > 0.f ? gv : 0.f;
      s0 += dz;
      s1 = __builtin_fmaf(dz, (xv - mu) * is, s1);
    }
  };
  if ((a.N & 3) == 0) {
    int it = 0;
    for (int n = n0 + 4 * (int)threadIdx.x; n < n1; n += 1024) {
      const float4 xv = *reinterpret_cast<const float4 *>(xr + n);
      float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (MODE == 1) gv = *reinterpret_cast<const float4 *>(gr + n);
      item(xv.x, gv.x); item(xv.y, gv.y); item(xv.z, gv.z); item(xv.w, gv.w);
      if (++it == 16) {
        d0 += (double)s0; d1 += (double)s1; s0 = s1 = 0.f; it = 0;
      }
    }
  } else {
    int it = 0;
    for (int n = n0 + (int)threadIdx.x; n < n1; n += 256) {
      item(xr[n], MODE == 1 ? gr[n] : 0.f);
      if (++it == 64) {
        d0 += (double)s0; d1 += (double)s1; s0 = s1 = 0.f; it = 0;
      }
    }
  }
  d0 += (double)s0;
  d1 += (double)s1;
  const double t0 = block_sum(d0, scratch);
  const double t1 = block_sum(d1, scratch);
  if (threadIdx.x == 0) bn_store_partial(a.partial + ((size_t)part * a.C + c) * 2, t0, t1, a.tickets != nullptr);
  if (a.tickets == nullptr) return;
  __shared__ int s_last;
  if (!last_arrival(a.tickets + c, gridDim.y, &s_last)) return;
  if (threadIdx.x < 64) bn_finalize_channel<MODE, true>(a.fin, c, threadIdx.x);
}

// MODE 0: out = ReLU(scale x + shift);  MODE 1: dx = A dz + Bc + D x with dz = g gated by the ReLU
template <int MODE>
__global__ __launch_bounds__(256) void bn_apply_kernel(BnArgs a) {
  const long long rows = (long long)a.B * a.C;
  const int per_row = (a.N + 1023) / 1024;  // 1024 elements per block-iteration
  for (long long t = blockIdx.x; t < rows * per_row; t += gridDim.x) {
    const long long r = t / per_row;
    const int c = (int)(r % a.C);
    const int n = (int)(t - r * per_row) * 1024 + 4 * (int)threadIdx.x;
    const float sc = a.scale[c], sh = a.shift[c];
    const float *xr = a.x + (size_t)r * a.N;
    float *orow = a.out + (size_t)r * a.N;
    float cA = 0.f, cB = 0.f, cD = 0.f;
    if (MODE == 1) {
      cA = a.cA[c]; cB = a.cB[c]; cD = a.cD[c];
    }
    auto f = [&](float xv, float gv) {
      const float z = __builtin_fmaf(xv, sc, sh);
      if (MODE == 0) return z > 0.f ? z : 0.f;
      const float dz = z > 0.f ? gv : 0.f;
      return __builtin_fmaf(cA, dz, __builtin_fmaf(cD, xv, cB));
    };
    if ((a.N & 3) == 0) {
      if (n < a.N) {
        const float4 xv = *reinterpret_cast<const float4 *>(xr + n);
        float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 1) gv = *reinterpret_cast<const float4 *>(a.g + (size_t)r * a.N + n);
        *reinterpret_cast<float4 *>(orow + n) = make_float4(f(xv.x, gv.x), f(xv.y, gv.y), f(xv.z, gv.z), f(xv.w, gv.w));
      }
    } else {
      for (int u = 0; u < 4; ++u)
        if (n + u < a.N) orow[n + u] = f(xr[n + u], MODE == 1 ? a.g[(size_t)r * a.N + n + u] : 0.f);
    }
  }
}

// the same algebra as a launch of its own (one 64-thread workgroup per channel): when no ticket piece is available
template <int MODE>
__global__ __launch_bounds__(64) void bn_finalize_kernel(BnFinArgs a) {
  bn_finalize_channel<MODE, false>(a, blockIdx.x, threadIdx.x);
}

// ---- the tail of a bottleneck: out = ReLU(BN1(x1) + R),  R = 0 | x2 (identity shortcut) | BN2(x2) (conv shortcut)
// (reference backbones/resnet.py:58-66: conv2's BatchNorm, the shortcut's BatchNorm, the add and the final ReLU --
// four element-wise passes and two BatchNorm kernels in the reference, one streaming pass here), and its backward:
// dz = g gated by the ReLU (mask read off the saved output), both BatchNorm backward reductions in ONE pass over
// (g, out, x1, x2), then dx1 = A1 dz + B1 + D1 x1 and dx2 = A2 dz + B2 + D2 x2 (or dx2 = dz) in one more.
struct Bn2Args {
  const float *x1, *x2, *g, *out_ref;
  const float *s1, *t1, *mu1, *is1, *s2, *t2, *mu2, *is2;
  const float *c1, *c2;  // coefficient blocks [5,C]: A, Bc, D, d gamma, d beta
  float *o1, *o2;
  double *p1, *p2;       // [B*chunks, C, 2] each
  int B, C, N, chunks, span;
  int mode2;             // 0: no second branch, 1: identity, 2: affine (its own BatchNorm)
  int relu;
  int vec;               // N % 4 == 0 and every tensor 16-byte aligned: the backward kernels move 16 bytes per lane
  unsigned *tickets;     // [C] or null; set: the last arrival at a channel's ticket runs fin1 (and fin2 when mode2 == 2)
  BnFinArgs fin1, fin2;  //   inside bn2_bwd_stats_kernel, no finalize launches follow
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

__global__ __launch_bounds__(256) void bn2_apply_kernel(Bn2Args a) {
  const long long rows = (long long)a.B * a.C;
  const int per_row = (a.N + 1023) / 1024;
  for (long long t = blockIdx.x; t < rows * per_row; t += gridDim.x) {
    const long long r = t / per_row;
    const int c = (int)(r % a.C);
    const int n = (int)(t - r * per_row) * 1024 + 4 * (int)threadIdx.x;
    const float s1 = a.s1[c], t1 = a.t1[c];
    const float s2 = a.mode2 == 2 ? a.s2[c] : 1.f, t2 = a.mode2 == 2 ? a.t2[c] : 0.f;
    const float *x1 = a.x1 + (size_t)r * a.N;
    const float *x2 = a.mode2 ? a.x2 + (size_t)r * a.N : nullptr;
    float *orow = a.o1 + (size_t)r * a.N;
    auto f = [&](float u, float v) {
      float z = __builtin_fmaf(u, s1, t1);
      if (a.mode2 == 1) z += v;
      else if (a.mode2 == 2) z += __builtin_fmaf(v, s2, t2);
      return (a.relu && !(z > 0.f)) ? 0.f : z;
    };
    if ((a.N & 3) == 0) {
      if (n < a.N) {
        const float4 u = *reinterpret_cast<const float4 *>(x1 + n);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.mode2) v = *reinterpret_cast<const float4 *>(x2 + n);
        *reinterpret_cast<float4 *>(orow + n) = make_float4(f(u.x, v.x), f(u.y, v.y), f(u.z, v.z), f(u.w, v.w));
      }
    } else {
      for (int e = 0; e < 4; ++e)
        if (n + e < a.N) orow[n + e] = f(x1[n + e], a.mode2 ? x2[n + e] : 0.f);
    }
  }
}

__global__ __launch_bounds__(256) void bn2_bwd_stats_kernel(Bn2Args a) {
  __shared__ double scratch[4];
  const int c = blockIdx.x, part = blockIdx.y;
  const int b = part / a.chunks, n0 = (part - b * a.chunks) * a.span;
  const int n1 = n0 + a.span < a.N ? n0 + a.span : a.N;
  const size_t row = ((size_t)b * a.C + c) * a.N;
  const float *x1 = a.x1 + row, *g = a.g + row;
  const float *x2 = a.mode2 == 2 ? a.x2 + row : nullptr;
  const float *ref = a.relu ? a.out_ref + row : nullptr;
  const float mu1 = a.mu1[c], is1 = a.is1[c];
  const float mu2 = a.mode2 == 2 ? a.mu2[c] : 0.f, is2 = a.mode2 == 2 ? a.is2[c] : 0.f;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;  // <= 64 terms per thread between the double folds
  double d0 = 0.0, d1 = 0.0, d2 = 0.0;
  int it = 0;
  if (a.vec) {
    // 16 bytes per lane and array (round 3; the scalar form below read 57 MB in 15.6 us at the operator benches'
    // [16,72,4096]: 3.6 TB/s)
    for (int n = n0 + 4 * (int)threadIdx.x; n < n1; n += 1024) {
      const float4 gv = ld4(g + n), xv = ld4(x1 + n);
      float4 rv = make_float4(1.f, 1.f, 1.f, 1.f), yv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.relu) rv = ld4(ref + n);
      if (a.mode2 == 2) yv = ld4(x2 + n);
      const float gg[4] = {gv.x, gv.y, gv.z, gv.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w};
      const float rr[4] = {rv.x, rv.y, rv.z, rv.w}, yy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dz = (a.relu && !(rr[e] > 0.f)) ? 0.f : gg[e];
        s0 += dz;
        s1 = __builtin_fmaf(dz, (xx[e] - mu1) * is1, s1);
        if (a.mode2 == 2) s2 = __builtin_fmaf(dz, (yy[e] - mu2) * is2, s2);
      }
      if (++it == 16) {
        d0 += (double)s0; d1 += (double)s1; d2 += (double)s2; s0 = s1 = s2 = 0.f; it = 0;
      }
    }
  } else {
    for (int n = n0 + (int)threadIdx.x; n < n1; n += 256) {
      const float dz = (a.relu && !(ref[n] > 0.f)) ? 0.f : g[n];
      s0 += dz;
      s1 = __builtin_fmaf(dz, (x1[n] - mu1) * is1, s1);
      if (a.mode2 == 2) s2 = __builtin_fmaf(dz, (x2[n] - mu2) * is2, s2);
      if (++it == 64) {
        d0 += (double)s0; d1 += (double)s1; d2 += (double)s2; s0 = s1 = s2 = 0.f; it = 0;
      }
    }
  }
  d0 += (double)s0; d1 += (double)s1; d2 += (double)s2;
  const double t0 = block_sum(d0, scratch);
  const double t1 = block_sum(d1, scratch);
  const double t2 = block_sum(d2, scratch);
  if (threadIdx.x == 0) {
    bn_store_partial(a.p1 + ((size_t)part * a.C + c) * 2, t0, t1, a.tickets != nullptr);
    if (a.mode2 == 2) bn_store_partial(a.p2 + ((size_t)part * a.C + c) * 2, t0, t2, a.tickets != nullptr);
  }
  if (a.tickets == nullptr) return;
  __shared__ int s_last;
  if (!last_arrival(a.tickets + c, gridDim.y, &s_last)) return;
  if (threadIdx.x < 64) bn_finalize_channel<1, true>(a.fin1, c, threadIdx.x);
  else if (threadIdx.x < 128 && a.mode2 == 2) bn_finalize_channel<1, true>(a.fin2, c, threadIdx.x - 64);
}

__global__ __launch_bounds__(256) void bn2_bwd_apply_kernel(Bn2Args a) {
  const long long rows = (long long)a.B * a.C;
  const int per_row = (a.N + 1023) / 1024;
  for (long long t = blockIdx.x; t < rows * per_row; t += gridDim.x) {
    const long long r = t / per_row;
    const int c = (int)(r % a.C);
    const int n = (int)(t - r * per_row) * 1024 + 4 * (int)threadIdx.x;
    const float A1 = a.c1[c], B1 = a.c1[a.C + c], D1 = a.c1[2 * a.C + c];
    float A2 = 1.f, B2 = 0.f, D2 = 0.f;
    if (a.mode2 == 2) {
      A2 = a.c2[c]; B2 = a.c2[a.C + c]; D2 = a.c2[2 * a.C + c];
    }
    const size_t row = (size_t)r * a.N;
    if (a.vec) {
      // every load of the item before its first store (the scalar form's stores, which may alias the inputs as far as
      // the compiler knows, serialised it: 57 MB in 23.6 us at [16,72,4096])
      if (n < a.N) {
        const float4 gv = ld4(a.g + row + n), xv = ld4(a.x1 + row + n);
        float4 rv = make_float4(1.f, 1.f, 1.f, 1.f), yv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.relu) rv = ld4(a.out_ref + row + n);
        if (a.mode2 == 2) yv = ld4(a.x2 + row + n);
        float4 dz;
        dz.x = (a.relu && !(rv.x > 0.f)) ? 0.f : gv.x;
        dz.y = (a.relu && !(rv.y > 0.f)) ? 0.f : gv.y;
        dz.z = (a.relu && !(rv.z > 0.f)) ? 0.f : gv.z;
        dz.w = (a.relu && !(rv.w > 0.f)) ? 0.f : gv.w;
        *reinterpret_cast<float4 *>(a.o1 + row + n) =
            make_float4(__builtin_fmaf(A1, dz.x, __builtin_fmaf(D1, xv.x, B1)), __builtin_fmaf(A1, dz.y, __builtin_fmaf(D1, xv.y, B1)),
                        __builtin_fmaf(A1, dz.z, __builtin_fmaf(D1, xv.z, B1)), __builtin_fmaf(A1, dz.w, __builtin_fmaf(D1, xv.w, B1)));
        if (a.mode2 == 1) *reinterpret_cast<float4 *>(a.o2 + row + n) = dz;
        else if (a.mode2 == 2)
          *reinterpret_cast<float4 *>(a.o2 + row + n) =
              make_float4(__builtin_fmaf(A2, dz.x, __builtin_fmaf(D2, yv.x, B2)), __builtin_fmaf(A2, dz.y, __builtin_fmaf(D2, yv.y, B2)),
                          __builtin_fmaf(A2, dz.z, __builtin_fmaf(D2, yv.z, B2)), __builtin_fmaf(A2, dz.w, __builtin_fmaf(D2, yv.w, B2)));
      }
      continue;
    }
    for (int e = 0; e < 4; ++e) {
      const int m = n + e;
      if (m >= a.N) break;
      const float dz = (a.relu && !(a.out_ref[row + m] > 0.f)) ? 0.f : a.g[row + m];
      a.o1[row + m] = __builtin_fmaf(A1, dz, __builtin_fmaf(D1, a.x1[row + m], B1));
      if (a.mode2 == 1) a.o2[row + m] = dz;
      else if (a.mode2 == 2) a.o2[row + m] = __builtin_fmaf(A2, dz, __builtin_fmaf(D2, a.x2[row + m], B2));
    }
  }
}

// ---- few values per channel (the deep stages: 16 clouds x 16 .. 1024 points, hundreds to thousands of channels): ONE
// workgroup per channel does the statistics pass, the per-channel algebra and the apply pass in a single launch (the
// second read comes out of L2) instead of three launches whose grids are mostly launch overhead.
constexpr int kBnSmallMax = 16384;  // B*N up to here (16 float4 items per thread and pass)

struct BnSmallArgs {
  const float *x1, *x2, *g, *out_ref;
  const float *gamma1, *beta1, *gamma2, *beta2;
  float *rm1, *rv1, *rm2, *rv2;
  long long *nbt1, *nbt2;  // num_batches_tracked of the two BatchNorms (may be null)
  float *vec1, *vec2;    // [4,C]: scale, shift, mean, invstd (written by forward, read by backward)
  float *coef1, *coef2;  // [5,C]: A, Bc, D, d gamma, d beta (backward)
  float *o1, *o2;
  int B, C, N;
  float eps1, mom1, eps2, mom2;
  int mode2, relu;
};

// walk one channel's B*N values, VEC (1 or 4) consecutive points at a time, four items in flight per thread;
// f(offset of the first value, how many of the VEC are valid) -- items never straddle a cloud (VEC == 4 needs N % 4 == 0)
template <int VEC, class F>
__device__ __forceinline__ void for_channel(int c, int B, int C, int N, F &&f) {
  const int per_row = N / VEC, items = B * per_row;
#pragma unroll 4
  for (int t = threadIdx.x; t < items; t += 256) {
    const int b = t / per_row, n = (t - b * per_row) * VEC;
    f(((size_t)b * C + c) * N + n);
  }
}

template <int VEC>
struct Pack {
  float v[VEC];
};
template <int VEC>
__device__ __forceinline__ Pack<VEC> ldp(const float *p) {
  Pack<VEC> r;
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
    r.v[0] = p[0];
  }
  return r;
}
template <int VEC>
__device__ __forceinline__ void stp(float *p, const Pack<VEC> &r) {
  if constexpr (VEC == 4) *reinterpret_cast<float4 *>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  else p[0] = r.v[0];
}

template <int VEC>
__device__ __forceinline__ void channel_stats(const float *x, int c, int B, int C, int N, double *scratch, double &mean,
                                              double &var) {
  float s0 = 0.f, s1 = 0.f;
  for_channel<VEC>(c, B, C, N, [&](size_t at) {
    const Pack<VEC> v = ldp<VEC>(x + at);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      s0 += v.v[e];
      s1 = __builtin_fmaf(v.v[e], v.v[e], s1);
    }
  });
  const int total = B * N;
  const double t0 = block_sum((double)s0, scratch);
  const double t1 = block_sum((double)s1, scratch);
  mean = t0 / total;
  var = t1 / total - mean * mean;
  var = var > 0.0 ? var : 0.0;
}

template <int VEC>
__global__ __launch_bounds__(256) void bn2_fwd_small_kernel(BnSmallArgs a) {
  __shared__ double scratch[4];
  __shared__ float s_par[4];
  const int c = blockIdx.x, total = a.B * a.N;
  double mean, var;
  channel_stats<VEC>(a.x1, c, a.B, a.C, a.N, scratch, mean, var);
  if (threadIdx.x == 0) {
    const double invstd = 1.0 / sqrt(var + (double)a.eps1), scale = (double)a.gamma1[c] * invstd;
    s_par[0] = (float)scale;
    s_par[1] = (float)((double)a.beta1[c] - mean * scale);
    a.vec1[c] = s_par[0]; a.vec1[a.C + c] = s_par[1]; a.vec1[2 * a.C + c] = (float)mean; a.vec1[3 * a.C + c] = (float)invstd;
    if (a.rm1) {
      const double unbiased = var * ((double)total / (total > 1 ? total - 1.0 : 1.0));
      a.rm1[c] = a.rm1[c] * (1.0f - a.mom1) + a.mom1 * (float)mean;
      a.rv1[c] = a.rv1[c] * (1.0f - a.mom1) + a.mom1 * (float)unbiased;
    }
    if (c == 0 && a.nbt1) *a.nbt1 += 1;
  }
  if (a.mode2 == 2) {
    channel_stats<VEC>(a.x2, c, a.B, a.C, a.N, scratch, mean, var);
    if (threadIdx.x == 0) {
      const double invstd = 1.0 / sqrt(var + (double)a.eps2), scale = (double)a.gamma2[c] * invstd;
      s_par[2] = (float)scale;
      s_par[3] = (float)((double)a.beta2[c] - mean * scale);
      a.vec2[c] = s_par[2]; a.vec2[a.C + c] = s_par[3]; a.vec2[2 * a.C + c] = (float)mean; a.vec2[3 * a.C + c] = (float)invstd;
      if (a.rm2) {
        const double unbiased = var * ((double)total / (total > 1 ? total - 1.0 : 1.0));
        a.rm2[c] = a.rm2[c] * (1.0f - a.mom2) + a.mom2 * (float)mean;
        a.rv2[c] = a.rv2[c] * (1.0f - a.mom2) + a.mom2 * (float)unbiased;
      }
      if (c == 0 && a.nbt2) *a.nbt2 += 1;
    }
  }
  __syncthreads();
  const float s1 = s_par[0], t1 = s_par[1];
  const float s2 = a.mode2 == 2 ? s_par[2] : 1.f, t2 = a.mode2 == 2 ? s_par[3] : 0.f;
  for_channel<VEC>(c, a.B, a.C, a.N, [&](size_t at) {
    const Pack<VEC> u = ldp<VEC>(a.x1 + at);
    Pack<VEC> v{}, o;
    if (a.mode2) v = ldp<VEC>(a.x2 + at);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float z = __builtin_fmaf(u.v[e], s1, t1);
      if (a.mode2 == 1) z += v.v[e];
      else if (a.mode2 == 2) z += __builtin_fmaf(v.v[e], s2, t2);
      o.v[e] = (a.relu && !(z > 0.f)) ? 0.f : z;
    }
    stp<VEC>(a.o1 + at, o);
  });
}

template <int VEC>
__global__ __launch_bounds__(256) void bn2_bwd_small_kernel(BnSmallArgs a) {
  __shared__ double scratch[4];
  __shared__ float s_co[6];
  const int c = blockIdx.x, total = a.B * a.N;
  const float mu1 = a.vec1[2 * a.C + c], is1 = a.vec1[3 * a.C + c];
  const float mu2 = a.mode2 == 2 ? a.vec2[2 * a.C + c] : 0.f, is2 = a.mode2 == 2 ? a.vec2[3 * a.C + c] : 0.f;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for_channel<VEC>(c, a.B, a.C, a.N, [&](size_t at) {
    const Pack<VEC> g = ldp<VEC>(a.g + at), u = ldp<VEC>(a.x1 + at);
    Pack<VEC> r{}, v{};
    if (a.relu) r = ldp<VEC>(a.out_ref + at);
    if (a.mode2 == 2) v = ldp<VEC>(a.x2 + at);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float dz = (a.relu && !(r.v[e] > 0.f)) ? 0.f : g.v[e];
      s0 += dz;
      s1 = __builtin_fmaf(dz, (u.v[e] - mu1) * is1, s1);
      if (a.mode2 == 2) s2 = __builtin_fmaf(dz, (v.v[e] - mu2) * is2, s2);
    }
  });
  const double t0 = block_sum((double)s0, scratch);
  const double t1 = block_sum((double)s1, scratch);
  const double t2 = block_sum((double)s2, scratch);
  if (threadIdx.x == 0) {
    auto coeffs = [&](double gamma, double mean, double invstd, double sxh, float *coef, float *sh) {
      const double A = gamma * invstd;
      const double D = -A * invstd * sxh / total;
      const double Bc = -A * t0 / total - D * mean;
      coef[c] = (float)A; coef[a.C + c] = (float)Bc; coef[2 * a.C + c] = (float)D;
      coef[3 * a.C + c] = (float)sxh; coef[4 * a.C + c] = (float)t0;
      sh[0] = (float)A; sh[1] = (float)Bc; sh[2] = (float)D;
    };
    coeffs((double)a.gamma1[c], (double)mu1, (double)is1, t1, a.coef1, s_co);
    if (a.mode2 == 2) coeffs((double)a.gamma2[c], (double)mu2, (double)is2, t2, a.coef2, s_co + 3);
  }
  __syncthreads();
  const float A1 = s_co[0], B1 = s_co[1], D1 = s_co[2];
  const float A2 = a.mode2 == 2 ? s_co[3] : 1.f, B2 = a.mode2 == 2 ? s_co[4] : 0.f, D2 = a.mode2 == 2 ? s_co[5] : 0.f;
  for_channel<VEC>(c, a.B, a.C, a.N, [&](size_t at) {
    const Pack<VEC> g = ldp<VEC>(a.g + at), u = ldp<VEC>(a.x1 + at);
    Pack<VEC> r{}, v{}, d1, d2;
    if (a.relu) r = ldp<VEC>(a.out_ref + at);
    if (a.mode2 == 2) v = ldp<VEC>(a.x2 + at);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float dz = (a.relu && !(r.v[e] > 0.f)) ? 0.f : g.v[e];
      d1.v[e] = __builtin_fmaf(A1, dz, __builtin_fmaf(D1, u.v[e], B1));
      d2.v[e] = a.mode2 == 2 ? __builtin_fmaf(A2, dz, __builtin_fmaf(D2, v.v[e], B2)) : dz;
    }
    stp<VEC>(a.o1 + at, d1);
    if (a.mode2) stp<VEC>(a.o2 + at, d2);
  });
}

// ---- BatchNorm + ReLU on POINT-MAJOR rows [P, C] (round 5: the PosPool / AdaptiveWeight / PseudoGrid bottlenecks) -------
// The fused reduction operators write their result as point-major rows, conv2 of the bottleneck reads rows and applies
// max(scale x + shift, 0) while it stages them (cl3d_conv1x1_rows_*): the operator's BatchNorm then needs only its
// STATISTICS forward (this kernel: column sums of a row-major matrix) and, backward, dx = A dz + Bc + D x on rows.  The
// channel-major pair above would need the rows transposed first -- the [B,C,N] round trip f1 removes.
// A workgroup owns `rows_per_block` consecutive rows: thread t -> 16-byte channel group t % CG of row offset t / CG
// (a wave covers whole 128-byte pieces of consecutive rows: coalesced); per-thread float sums folded to double every
// 16 rows; the row offsets are added in order through LDS.  Partials [G, C, 2] as the channel-major kernels write them,
// so the same finalize kernels serve.
struct BnRowsArgs {
  const float *x, *g;      // [P, C]
  const float *scale, *shift, *mean, *invstd, *cA, *cB, *cD;
  float *out;              // [P, C]
  double *partial;         // [G, C, 2]
  long long P;
  int C, rows_per_block;
};

template <int MODE>
__global__ __launch_bounds__(256) void bn_rows_stats_kernel(BnRowsArgs a) {
  extern __shared__ double rs_lds[];  // [RO][CG][8]
  const int CG = a.C >> 2;            // 16-byte channel groups per row
  const int cgs = CG < 256 ? CG : 256;
  const int RO = 256 / cgs;           // row offsets handled side by side
  const int cgl = (int)threadIdx.x % cgs, ro = (int)threadIdx.x / cgs;
  const long long p0 = (long long)blockIdx.x * a.rows_per_block;
  const long long p1 = p0 + a.rows_per_block < a.P ? p0 + a.rows_per_block : a.P;
  for (int cg0 = 0; cg0 < CG; cg0 += cgs) {  // (one pass for C <= 1024; the trip count is the same for every thread)
    const int cg = cg0 + cgl;
    const bool live = cg < CG;
    const int c = 4 * (live ? cg : 0);
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc, mu = sc, is = sc;
    if (MODE == 1) {
      sc = *reinterpret_cast<const float4 *>(a.scale + c); sh = *reinterpret_cast<const float4 *>(a.shift + c);
      mu = *reinterpret_cast<const float4 *>(a.mean + c); is = *reinterpret_cast<const float4 *>(a.invstd + c);
    }
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    double d0[4] = {0.0, 0.0, 0.0, 0.0}, d1[4] = {0.0, 0.0, 0.0, 0.0};
    int it = 0;
    if (ro < RO && live) {
      for (long long p = p0 + ro; p < p1; p += RO) {
        const float4 xv = *reinterpret_cast<const float4 *>(a.x + (size_t)p * a.C + c);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0[e] += xs[e];
            s1[e] = __builtin_fmaf(xs[e], xs[e], s1[e]);
          }
        } else {
          const float4 gv = *reinterpret_cast<const float4 *>(a.g + (size_t)p * a.C + c);
          const float gs[4] = {gv.x, gv.y, gv.z, gv.w};
          const float scs[4] = {sc.x, sc.y, sc.z, sc.w}, shs[4] = {sh.x, sh.y, sh.z, sh.w};
          const float mus[4] = {mu.x, mu.y, mu.z, mu.w}, iss[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dz = __builtin_fmaf(xs[e], scs[e], shs[e]) > 0.f ? gs[e] : 0.f;
            s0[e] += dz;
            s1[e] = __builtin_fmaf(dz, (xs[e] - mus[e]) * iss[e], s1[e]);
          }
        }
        if (++it == 16) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            d0[e] += (double)s0[e]; d1[e] += (double)s1[e]; s0[e] = s1[e] = 0.f;
          }
          it = 0;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d0[e] += (double)s0[e];
      d1[e] += (double)s1[e];
    }
    __syncthreads();  // (the previous channel pass has been read out)
    if (ro < RO) {
      double *l = rs_lds + ((size_t)ro * cgs + cgl) * 8;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        l[2 * e] = d0[e];
        l[2 * e + 1] = d1[e];
      }
    }
    __syncthreads();
    if (ro == 0 && live) {
      double t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = rs_lds[(size_t)cgl * 8 + e];
      for (int r = 1; r < RO; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] += rs_lds[((size_t)r * cgs + cgl) * 8 + e];
      double *pp = a.partial + ((size_t)blockIdx.x * a.C + c) * 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) pp[e] = t[e];  // [c .. c+3][2]
    }
  }
}

// dx = A dz + Bc + D x on rows, dz = g gated by the ReLU recomputed from x
__global__ __launch_bounds__(256) void bn_rows_bwd_apply_kernel(BnRowsArgs a) {
  const int CG = a.C >> 2;
  const long long total = a.P * CG;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int c = 4 * (int)(t % CG);
    const float4 xv = *reinterpret_cast<const float4 *>(a.x + t * 4), gv = *reinterpret_cast<const float4 *>(a.g + t * 4);
    const float4 sc = *reinterpret_cast<const float4 *>(a.scale + c), sh = *reinterpret_cast<const float4 *>(a.shift + c);
    const float4 cA = *reinterpret_cast<const float4 *>(a.cA + c), cB = *reinterpret_cast<const float4 *>(a.cB + c);
    const float4 cD = *reinterpret_cast<const float4 *>(a.cD + c);
    auto f = [](float x, float g, float s, float h, float A, float B, float D) {
      const float dz = __builtin_fmaf(x, s, h) > 0.f ? g : 0.f;
      return __builtin_fmaf(A, dz, __builtin_fmaf(D, x, B));
    };
    *reinterpret_cast<float4 *>(a.out + t * 4) =
        make_float4(f(xv.x, gv.x, sc.x, sh.x, cA.x, cB.x, cD.x), f(xv.y, gv.y, sc.y, sh.y, cA.y, cB.y, cD.y),
                    f(xv.z, gv.z, sc.z, sh.z, cA.z, cB.z, cD.z), f(xv.w, gv.w, sc.w, sh.w, cA.w, cB.w, cD.w));
  }
}

static int bn_rows_block(long long P) {  // rows per workgroup: ~1024 workgroups, at least 32 rows each
  long long r = (P + 1023) / 1024;
  r = r < 32 ? 32 : r;
  return (int)((r + 7) & ~7LL);
}

static void bn_shape(BnArgs &a) {
  a.span = kBnSpan;
  a.chunks = ceil_div(a.N, a.span);
}

// One ticket per channel for a statistics launch whose last-arriving workgroup finishes the channel itself (round 6; one
// launch and one graph node fewer per BatchNorm pass).  CL3D_BN_FOLD=0 builds the form with a finalize launch of its own
// (the A/B arm of scripts/micro/kernel_variants.py); a null piece (ring used up by captured launches, first call inside
// a capture) takes that form too.
#ifndef CL3D_BN_FOLD
#define CL3D_BN_FOLD 1
#endif
static unsigned *bn_tickets(int C, hipStream_t st) {
  if (!CL3D_BN_FOLD) return nullptr;
  return ticket_piece((size_t)C, st);
}

}  // namespace cl3d

extern "C" int cl3d_bn_partials(int B, int C, int N) {
  (void)C;
  return B * cl3d::ceil_div(N > 0 ? N : 1, cl3d::kBnSpan);
}

extern "C" int cl3d_bn_relu_stats(const float *x, int B, int C, int N, double *partial, int n_partials, double count,
                                  float eps, float momentum, const float *gamma, const float *beta,
                                  float *running_mean, float *running_var, int64_t *num_batches_tracked, float *scale,
                                  float *shift, float *mean, float *invstd, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 1 && C >= 1 && N >= 1 && count > 0, "bn_relu_stats: bad sizes");
  CL3D_REQUIRE(x && partial && gamma && beta && scale && shift && mean && invstd, "bn_relu_stats: null pointer");
  CL3D_REQUIRE(n_partials == cl3d_bn_partials(B, C, N) && n_partials <= 65535, "bn_relu_stats: wrong partial count");
  BnArgs a{};
  a.x = x; a.partial = partial; a.B = B; a.C = C; a.N = N;
  bn_shape(a);
  BnFinArgs f{};
  f.partial = partial; f.G = n_partials; f.C = C; f.count = count; f.eps = eps; f.momentum = momentum;
  f.gamma = gamma; f.beta = beta; f.running_mean = running_mean; f.running_var = running_var;
  f.num_batches_tracked = reinterpret_cast<long long *>(num_batches_tracked);
  f.o0 = scale; f.o1 = shift; f.o2 = mean; f.o3 = invstd;
  a.tickets = bn_tickets(C, (hipStream_t)stream);
  a.fin = f;
  hipLaunchKernelGGL((bn_stats_kernel<0>), dim3(C, n_partials), dim3(256), 0, (hipStream_t)stream, a);
  if (!a.tickets) hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3(C), dim3(64), 0, (hipStream_t)stream, f);
  return check_launch("cl3d_bn_relu_stats");
}

extern "C" int cl3d_bn_relu_apply(const float *x, const float *scale, const float *shift, int B, int C, int N,
                                  float *out, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 0 && C >= 1 && N >= 0, "bn_relu_apply: bad sizes");
  if (B == 0 || N == 0) return CL3D_OK;
  CL3D_REQUIRE(x && scale && shift && out, "bn_relu_apply: null pointer");
  BnArgs a{};
  a.x = x; a.scale = scale; a.shift = shift; a.out = out; a.B = B; a.C = C; a.N = N;
  const long long work = (long long)B * C * ceil_div(N, 1024);
  hipLaunchKernelGGL((bn_apply_kernel<0>), dim3((unsigned)(work < 65536 ? work : 65536)), dim3(256), 0,
                     (hipStream_t)stream, a);
  return check_launch("cl3d_bn_relu_apply");
}

extern "C" int cl3d_bn_relu_bwd(const float *g, const float *x, const float *scale, const float *shift,
                                const float *mean, const float *invstd, const float *gamma, int B, int C, int N,
                                double count, double *partial, int n_partials, float *coef, float *dx,
                                cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 1 && C >= 1 && N >= 1 && count > 0, "bn_relu_bwd: bad sizes");
  CL3D_REQUIRE(g && x && scale && shift && mean && invstd && gamma && partial && coef && dx, "bn_relu_bwd: null pointer");
  CL3D_REQUIRE(n_partials == cl3d_bn_partials(B, C, N) && n_partials <= 65535, "bn_relu_bwd: wrong partial count");
  BnArgs a{};
  a.x = x; a.g = g; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.partial = partial;
  a.B = B; a.C = C; a.N = N;
  bn_shape(a);
  BnFinArgs f{};
  f.partial = partial; f.G = n_partials; f.C = C; f.count = count; f.gamma = gamma; f.mean_in = mean; f.invstd_in = invstd;
  f.o0 = coef; f.o1 = coef + C; f.o2 = coef + 2 * C; f.o3 = coef + 3 * C; f.o4 = coef + 4 * C;  // A, Bc, D, d gamma, d beta
  a.tickets = bn_tickets(C, (hipStream_t)stream);
  a.fin = f;
  hipLaunchKernelGGL((bn_stats_kernel<1>), dim3(C, n_partials), dim3(256), 0, (hipStream_t)stream, a);
  if (!a.tickets) hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3(C), dim3(64), 0, (hipStream_t)stream, f);
  a.tickets = nullptr;
  a.cA = coef; a.cB = coef + C; a.cD = coef + 2 * C; a.out = dx;
  const long long work = (long long)B * C * ceil_div(N, 1024);
  hipLaunchKernelGGL((bn_apply_kernel<1>), dim3((unsigned)(work < 65536 ? work : 65536)), dim3(256), 0,
                     (hipStream_t)stream, a);
  return check_launch("cl3d_bn_relu_bwd");
}

extern "C" int cl3d_bn_add_relu_apply(const float *x1, const float *scale1, const float *shift1, const float *x2,
                                      const float *scale2, const float *shift2, int relu, int B, int C, int N,
                                      float *out, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 0 && C >= 1 && N >= 0, "bn_add_relu_apply: bad sizes");
  if (B == 0 || N == 0) return CL3D_OK;
  CL3D_REQUIRE(x1 && scale1 && shift1 && out && (!scale2 || (x2 && shift2)), "bn_add_relu_apply: null pointer");
  Bn2Args a{};
  a.x1 = x1; a.s1 = scale1; a.t1 = shift1; a.x2 = x2; a.s2 = scale2; a.t2 = shift2; a.o1 = out;
  a.B = B; a.C = C; a.N = N; a.relu = relu; a.mode2 = !x2 ? 0 : (scale2 ? 2 : 1);
  const long long work = (long long)B * C * ceil_div(N, 1024);
  hipLaunchKernelGGL(bn2_apply_kernel, dim3((unsigned)(work < 65536 ? work : 65536)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("cl3d_bn_add_relu_apply");
}

extern "C" int cl3d_bn_add_relu_bwd(const float *g, const float *out, const float *x1, const float *mean1,
                                    const float *invstd1, const float *gamma1, const float *x2, const float *mean2,
                                    const float *invstd2, const float *gamma2, int relu, int B, int C, int N,
                                    double count, double *partial, int n_partials, float *coef1, float *coef2,
                                    float *dx1, float *dx2, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 1 && C >= 1 && N >= 1 && count > 0, "bn_add_relu_bwd: bad sizes");
  CL3D_REQUIRE(g && x1 && mean1 && invstd1 && gamma1 && partial && coef1 && dx1 && (!relu || out), "bn_add_relu_bwd: null pointer");
  CL3D_REQUIRE(!x2 || dx2, "bn_add_relu_bwd: second branch needs its gradient buffer");
  CL3D_REQUIRE(!gamma2 || (x2 && mean2 && invstd2 && coef2), "bn_add_relu_bwd: second BatchNorm incomplete");
  CL3D_REQUIRE(n_partials == cl3d_bn_partials(B, C, N) && n_partials <= 65535, "bn_add_relu_bwd: wrong partial count");
  hipStream_t st = (hipStream_t)stream;
  if ((long long)B * N <= kBnSmallMax && mean1 + C == invstd1 && (!gamma2 || mean2 + C == invstd2)) {
    // few values per channel and the statistics in the [4,C] block cl3d_bn_add_relu_train_fwd leaves: one launch
    BnSmallArgs s{};
    s.g = g; s.out_ref = out; s.x1 = x1; s.x2 = x2; s.gamma1 = gamma1; s.gamma2 = gamma2;
    s.vec1 = const_cast<float *>(mean1) - 2 * C; s.vec2 = gamma2 ? const_cast<float *>(mean2) - 2 * C : nullptr;
    s.coef1 = coef1; s.coef2 = coef2; s.o1 = dx1; s.o2 = dx2; s.B = B; s.C = C; s.N = N;
    s.mode2 = !x2 ? 0 : (gamma2 ? 2 : 1); s.relu = relu;
    if ((N & 3) == 0) hipLaunchKernelGGL(bn2_bwd_small_kernel<4>, dim3(C), dim3(256), 0, st, s);
    else hipLaunchKernelGGL(bn2_bwd_small_kernel<1>, dim3(C), dim3(256), 0, st, s);
    return check_launch("cl3d_bn_add_relu_bwd(small)");
  }
  Bn2Args a{};
  a.g = g; a.out_ref = out; a.x1 = x1; a.mu1 = mean1; a.is1 = invstd1; a.x2 = x2; a.mu2 = mean2; a.is2 = invstd2;
  a.B = B; a.C = C; a.N = N; a.relu = relu; a.mode2 = !x2 ? 0 : (gamma2 ? 2 : 1);
  a.p1 = partial; a.p2 = partial + (size_t)n_partials * C * 2;
  a.span = kBnSpan;
  a.chunks = ceil_div(N, a.span);
  a.vec = (N & 3) == 0 && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(x1) |
                            reinterpret_cast<uintptr_t>(x2) | reinterpret_cast<uintptr_t>(dx1) | reinterpret_cast<uintptr_t>(dx2)) & 15u) == 0;
  BnFinArgs f{};
  f.partial = a.p1; f.G = n_partials; f.C = C; f.count = count; f.gamma = gamma1; f.mean_in = mean1; f.invstd_in = invstd1;
  f.o0 = coef1; f.o1 = coef1 + C; f.o2 = coef1 + 2 * C; f.o3 = coef1 + 3 * C; f.o4 = coef1 + 4 * C;
  a.fin1 = f;
  if (a.mode2 == 2) {
    f.partial = a.p2; f.gamma = gamma2; f.mean_in = mean2; f.invstd_in = invstd2;
    f.o0 = coef2; f.o1 = coef2 + C; f.o2 = coef2 + 2 * C; f.o3 = coef2 + 3 * C; f.o4 = coef2 + 4 * C;
    a.fin2 = f;
  }
  a.tickets = bn_tickets(C, st);
  hipLaunchKernelGGL(bn2_bwd_stats_kernel, dim3(C, n_partials), dim3(256), 0, st, a);
  if (!a.tickets) {
    hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3(C), dim3(64), 0, st, a.fin1);
    if (a.mode2 == 2) hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3(C), dim3(64), 0, st, a.fin2);
  }
  a.c1 = coef1; a.c2 = coef2; a.o1 = dx1; a.o2 = dx2;
  const long long work = (long long)B * C * ceil_div(N, 1024);
  hipLaunchKernelGGL(bn2_bwd_apply_kernel, dim3((unsigned)(work < 65536 ? work : 65536)), dim3(256), 0, st, a);
  return check_launch("cl3d_bn_add_relu_bwd");
}

// training forward in one call: batch statistics of x1 (and of x2 when it has its own BatchNorm: gamma2 != NULL), running
// statistics updated with nn.BatchNorm1d's rule, vec1 / vec2 [4,C] = scale, shift, mean, invstd left for the backward
// pass, out = act(BN1(x1) + R).  One launch when a channel has few values, statistics + apply passes otherwise.
extern "C" int cl3d_bn_add_relu_train_fwd(const float *x1, const float *gamma1, const float *beta1, float *running_mean1,
                                          float *running_var1, int64_t *num_batches_tracked1, float eps1, float momentum1,
                                          const float *x2, const float *gamma2, const float *beta2, float *running_mean2,
                                          float *running_var2, int64_t *num_batches_tracked2, float eps2, float momentum2,
                                          int relu, int B, int C, int N,
                                          double *partial, int n_partials, float *vec1, float *vec2, float *out,
                                          cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 1 && C >= 1 && N >= 1, "bn_add_relu_train_fwd: bad sizes");
  CL3D_REQUIRE(x1 && gamma1 && beta1 && vec1 && out && (!gamma2 || (x2 && beta2 && vec2)), "bn_add_relu_train_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int mode2 = !x2 ? 0 : (gamma2 ? 2 : 1);
  if ((long long)B * N <= kBnSmallMax) {
    BnSmallArgs s{};
    s.x1 = x1; s.x2 = x2; s.gamma1 = gamma1; s.beta1 = beta1; s.gamma2 = gamma2; s.beta2 = beta2;
    s.rm1 = running_mean1; s.rv1 = running_var1; s.rm2 = running_mean2; s.rv2 = running_var2;
    s.nbt1 = reinterpret_cast<long long *>(num_batches_tracked1);
    s.nbt2 = reinterpret_cast<long long *>(num_batches_tracked2);
    s.vec1 = vec1; s.vec2 = vec2; s.o1 = out; s.B = B; s.C = C; s.N = N;
    s.eps1 = eps1; s.mom1 = momentum1; s.eps2 = eps2; s.mom2 = momentum2; s.mode2 = mode2; s.relu = relu;
    if ((N & 3) == 0) hipLaunchKernelGGL(bn2_fwd_small_kernel<4>, dim3(C), dim3(256), 0, st, s);
    else hipLaunchKernelGGL(bn2_fwd_small_kernel<1>, dim3(C), dim3(256), 0, st, s);
    return check_launch("cl3d_bn_add_relu_train_fwd(small)");
  }
  CL3D_REQUIRE(partial && n_partials == cl3d_bn_partials(B, C, N), "bn_add_relu_train_fwd: partial buffer");
  int rc = cl3d_bn_relu_stats(x1, B, C, N, partial, n_partials, (double)B * N, eps1, momentum1, gamma1, beta1, running_mean1,
                              running_var1, num_batches_tracked1, vec1, vec1 + C, vec1 + 2 * C, vec1 + 3 * C, stream);
  if (rc != CL3D_OK) return rc;
  if (mode2 == 2) {
    rc = cl3d_bn_relu_stats(x2, B, C, N, partial, n_partials, (double)B * N, eps2, momentum2, gamma2, beta2, running_mean2,
                            running_var2, num_batches_tracked2, vec2, vec2 + C, vec2 + 2 * C, vec2 + 3 * C, stream);
    if (rc != CL3D_OK) return rc;
  }
  return cl3d_bn_add_relu_apply(x1, vec1, vec1 + C, x2, mode2 == 2 ? vec2 : nullptr, mode2 == 2 ? vec2 + C : nullptr, relu, B,
                                C, N, out, stream);
}

// ---- point-major rows [P, C] (C % 4 == 0): statistics forward, dx = A dz + Bc + D x backward -------------------------
extern "C" int cl3d_bn_rows_partials(long long P, int C) {
  (void)C;
  if (P <= 0) return 1;
  const int rpb = cl3d::bn_rows_block(P);
  return (int)((P + rpb - 1) / rpb);
}

static int bn_rows_checks(const float *x, long long P, int C, int n_partials, const char *who) {
  using namespace cl3d;
  if (P < 1 || C < 4 || (C & 3) != 0) return fail(CL3D_E_INVALID, "%s: needs P >= 1 and C a positive multiple of 4", who);
  if ((reinterpret_cast<uintptr_t>(x) & 15u) != 0) return fail(CL3D_E_INVALID, "%s: rows must be 16-byte aligned", who);
  if (n_partials != cl3d_bn_rows_partials(P, C)) return fail(CL3D_E_INVALID, "%s: wrong partial count", who);
  return CL3D_OK;
}

static size_t bn_rows_lds(int C) {
  const int CG = C >> 2, cgs = CG < 256 ? CG : 256;
  return (size_t)(256 / cgs) * cgs * 8 * sizeof(double);
}

extern "C" int cl3d_bn_rows_stats(const float *rows, long long P, int C, double *partial, int n_partials, double count,
                                  float eps, float momentum, const float *gamma, const float *beta, float *running_mean,
                                  float *running_var, int64_t *num_batches_tracked, float *scale, float *shift,
                                  float *mean, float *invstd, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(rows && partial && gamma && beta && scale && shift && mean && invstd && count > 0, "bn_rows_stats: null pointer");
  int rc = bn_rows_checks(rows, P, C, n_partials, "bn_rows_stats");
  if (rc != CL3D_OK) return rc;
  BnRowsArgs a{};
  a.x = rows; a.partial = partial; a.P = P; a.C = C; a.rows_per_block = bn_rows_block(P);
  hipLaunchKernelGGL((bn_rows_stats_kernel<0>), dim3(n_partials), dim3(256), bn_rows_lds(C), (hipStream_t)stream, a);
  BnFinArgs f{};
  f.partial = partial; f.G = n_partials; f.C = C; f.count = count; f.eps = eps; f.momentum = momentum;
  f.gamma = gamma; f.beta = beta; f.running_mean = running_mean; f.running_var = running_var;
  f.num_batches_tracked = reinterpret_cast<long long *>(num_batches_tracked);
  f.o0 = scale; f.o1 = shift; f.o2 = mean; f.o3 = invstd;
  hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3(C), dim3(64), 0, (hipStream_t)stream, f);
  return check_launch("cl3d_bn_rows_stats");
}

// g = gradient with respect to the ACTIVATED rows max(scale x + shift, 0); coef [5, C] = A, Bc, D, d gamma, d beta
extern "C" int cl3d_bn_rows_bwd(const float *g, const float *rows, const float *scale, const float *shift,
                                const float *mean, const float *invstd, const float *gamma, long long P, int C,
                                double count, double *partial, int n_partials, float *coef, float *drows,
                                cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(g && rows && scale && shift && mean && invstd && gamma && partial && coef && drows && count > 0,
               "bn_rows_bwd: null pointer");
  int rc = bn_rows_checks(rows, P, C, n_partials, "bn_rows_bwd");
  if (rc != CL3D_OK) return rc;
  CL3D_REQUIRE(((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(drows)) & 15u) == 0, "bn_rows_bwd: unaligned");
  BnRowsArgs a{};
  a.x = rows; a.g = g; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.partial = partial;
  a.P = P; a.C = C; a.rows_per_block = bn_rows_block(P);
  hipLaunchKernelGGL((bn_rows_stats_kernel<1>), dim3(n_partials), dim3(256), bn_rows_lds(C), (hipStream_t)stream, a);
  BnFinArgs f{};
  f.partial = partial; f.G = n_partials; f.C = C; f.count = count; f.gamma = gamma; f.mean_in = mean; f.invstd_in = invstd;
  f.o0 = coef; f.o1 = coef + C; f.o2 = coef + 2 * C; f.o3 = coef + 3 * C; f.o4 = coef + 4 * C;
  hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3(C), dim3(64), 0, (hipStream_t)stream, f);
  a.cA = coef; a.cB = coef + C; a.cD = coef + 2 * C; a.out = drows;
  const long long work = (P * (C >> 2) + 255) / 256;
  hipLaunchKernelGGL(bn_rows_bwd_apply_kernel, dim3((unsigned)(work < 16384 ? work : 16384)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("cl3d_bn_rows_bwd");
}

// group.hip -- neighbour gather (forward) and scatter-add (backward) for gfx950.
//
// Replaces group_points_gpu.cu:13-33 (gather) and :48-69 (float atomicAdd scatter) of the
// reference: one block per cloud, every thread re-reading the K indices for every channel,
// stores strided by K.  Here:
//
//   * forward: one block owns one (cloud, channel) feature row [N] and a slice of the M*K output
//     row.  The feature row is staged once into LDS (N*4 bytes; 16 KiB at N=4096) so the random
//     reads hit LDS banks (~8 lanes/clk/CU under random conflicts) instead of the vector L1
//     (1 lane/clk/CU for divergent addresses); the index stream is read as int4 and the output
//     is written as float4 -- both fully coalesced, 1 KiB per wave instruction.  HBM traffic is
//     the algorithmic minimum: 4*C*M*K bytes out, 4*C*N in (+ the index stream from L2).
//   * backward: the mirror image.  One block owns one (cloud, channel) gradient row; its waves stream
//     grad_out (float4, coalesced, prefetched) and accumulate into one LDS row of doubles
//     (ds_add_f64), rounded to float once at the end.  No global atomics.
//   * rows that do not fit LDS (N > kMaxLdsRow = 16384): the forward takes the direct
//     global-memory gather below; the backward tiles the support range, one LDS tile per block.
#include "cl3d_common.h"

namespace cl3d {

constexpr int kMaxLdsRow = 16384;  // floats: 64 KiB, the no-opt-in dynamic LDS limit
constexpr int kUnroll = 4;

// The gathered tensor is written once and never re-read by this kernel: a non-temporal store keeps the
// 537 MB stream from evicting the index stream and feature rows other workgroups are reading from L2.
__device__ __forceinline__ void store_stream(float4 *p, const float4 &v) {
  __builtin_nontemporal_store(v.x, &p->x);
  __builtin_nontemporal_store(v.y, &p->y);
  __builtin_nontemporal_store(v.z, &p->z);
  __builtin_nontemporal_store(v.w, &p->w);
}

__device__ __forceinline__ float4 load_stream(const float4 *p) {  // read-once stream (grad_out)
  float4 v;
  v.x = __builtin_nontemporal_load(&p->x);
  v.y = __builtin_nontemporal_load(&p->y);
  v.z = __builtin_nontemporal_load(&p->z);
  v.w = __builtin_nontemporal_load(&p->w);
  return v;
}

// ------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void group_fwd_lds_kernel(const float *__restrict__ points,
                                                            const int *__restrict__ idx, int C,
                                                            int N, int MK, int chunk,
                                                            float *__restrict__ out) {
  extern __shared__ float row[];
  const int bc = blockIdx.y;  // b*C + c
  const int b = bc / C;
  const float *src = points + (size_t)bc * N;
  for (int i = threadIdx.x; i < N; i += 256) row[i] = src[i];
  __syncthreads();

  const int *ib = idx + (size_t)b * MK;
  float *ob = out + (size_t)bc * MK;
  const int e0 = blockIdx.x * chunk;
  int e1 = e0 + chunk;
  e1 = e1 < MK ? e1 : MK;
  if (((MK | chunk) & 3) == 0) {
    const int4 *i4 = reinterpret_cast<const int4 *>(ib);
    float4 *o4 = reinterpret_cast<float4 *>(ob);
    const int g1 = e1 >> 2;
    int g = (e0 >> 2) + threadIdx.x;
    // kUnroll index loads in flight per thread before the first LDS read: the loop is otherwise
    // bound by the latency of one idx load per iteration
    for (; g + (kUnroll - 1) * 256 < g1; g += kUnroll * 256) {
      int4 ii[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) ii[u] = i4[g + u * 256];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        float4 v;
        v.x = row[ii[u].x];
        v.y = row[ii[u].y];
        v.z = row[ii[u].z];
        v.w = row[ii[u].w];
        store_stream(&o4[g + u * 256], v);
      }
    }
    for (; g < g1; g += 256) {
      const int4 ii = i4[g];
      float4 v;
      v.x = row[ii.x];
      v.y = row[ii.y];
      v.z = row[ii.z];
      v.w = row[ii.w];
      store_stream(&o4[g], v);
    }
  } else {
    for (int e = e0 + threadIdx.x; e < e1; e += 256) ob[e] = row[ib[e]];
  }
}

__global__ __launch_bounds__(256) void group_fwd_direct_kernel(const float *__restrict__ points,
                                                               const int *__restrict__ idx, int C,
                                                               int N, int MK,
                                                               float *__restrict__ out) {
  const int bc = blockIdx.y;
  const int b = bc / C;
  const float *src = points + (size_t)bc * N;
  const int *ib = idx + (size_t)b * MK;
  float *ob = out + (size_t)bc * MK;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < MK; e += gridDim.x * 256) ob[e] = src[ib[e]];
}

// ----------------------------------------------------------------------------- backward
// One block owns one (cloud, channel) gradient row and a tile of T support indices; its four waves
// stream disjoint slices of grad_out (float4 + int4, software-prefetched one stage ahead) and
// accumulate into ONE shared LDS row of doubles with ds_add_f64.
// Why f64: measured on gfx950 (scripts/microbench_lds.hip, random addresses, op/clk/CU):
//   ds_add_f32 0.33 | ds_add_f64 2.5 | ds_add_u32 8.1 | plain read-add-write 4.3 | tag-resolved RMW 2.8.
// The f32 LDS atomic is 8x slower than the f64 one.  A double accumulator also makes the result
// independent of the order the waves arrive in, up to the final rounding to float: it equals
// round_f32(exact sum) except when the exact sum lies within ~1e-15 relative of a rounding boundary.
// A wave-private tag-resolved float row is strictly order-fixed, but at 8 B per accumulator it allows
// only one wave per 32 KiB of LDS and was measured latency-bound (0.60 ms vs 0.67 ms for ds_add_f32).
constexpr int kBwdStage = 2;  // float4 groups per lane per pipeline stage

__device__ __forceinline__ void add4(double *acc, int n0, unsigned span, const int4 &ii, const float4 &v) {
  if ((unsigned)(ii.x - n0) < span) atomicAdd(&acc[ii.x - n0], (double)v.x);
  if ((unsigned)(ii.y - n0) < span) atomicAdd(&acc[ii.y - n0], (double)v.y);
  if ((unsigned)(ii.z - n0) < span) atomicAdd(&acc[ii.z - n0], (double)v.z);
  if ((unsigned)(ii.w - n0) < span) atomicAdd(&acc[ii.w - n0], (double)v.w);
}

// CH channel rows of one cloud per block share every index load (the cloud's index stream is re-read once per
// block: at C = 64 that is as many L2 bytes as the gradient itself when CH = 1).
template <int CH>
__global__ __launch_bounds__(256) void group_bwd_lds_kernel(const float *__restrict__ grad_out,
                                                            const int *__restrict__ idx, int C, int N,
                                                            int MK, int T,
                                                            float *__restrict__ grad_points) {
  extern __shared__ double acc[];  // [CH][T]: this block owns support indices [n0, n0+T) of CH channels
  const int bc = blockIdx.x * CH;  // first (cloud, channel) row; C % CH == 0, so all CH rows are of one cloud
  const int b = bc / C;
  const int n0 = blockIdx.y * T;
  const unsigned span = (unsigned)(N - n0 < T ? N - n0 : T);
  for (int i = threadIdx.x; i < CH * T; i += 256) acc[i] = 0.0;
  __syncthreads();

  const float *g = grad_out + (size_t)bc * MK;
  const int *ib = idx + (size_t)b * MK;
  if ((MK & 3) == 0) {
    const int4 *i4 = reinterpret_cast<const int4 *>(ib);
    const int nq = MK >> 2;
    constexpr int kStride = 256 * kBwdStage;
    int4 ci[kBwdStage], ni[kBwdStage];
    float4 cv[kBwdStage][CH], nv[kBwdStage][CH];
    const int4 none = make_int4(-1, -1, -1, -1);
    int q = threadIdx.x;
#pragma unroll
    for (int u = 0; u < kBwdStage; ++u) {
      const int qq = q + u * 256;
      const int qc = qq < nq ? qq : nq - 1;  // always a valid address; out-of-range lanes are disabled via idx
      ci[u] = i4[qc];
#pragma unroll
      for (int c = 0; c < CH; ++c) cv[u][c] = load_stream(reinterpret_cast<const float4 *>(g + (size_t)c * MK) + qc);
      if (qq >= nq) ci[u] = none;
    }
    for (; q < nq; q += kStride) {
#pragma unroll
      for (int u = 0; u < kBwdStage; ++u) {  // next stage's loads are in flight while this one is summed
        const int qq = q + kStride + u * 256;
        const int qc = qq < nq ? qq : nq - 1;
        ni[u] = i4[qc];
#pragma unroll
        for (int c = 0; c < CH; ++c) nv[u][c] = load_stream(reinterpret_cast<const float4 *>(g + (size_t)c * MK) + qc);
        if (qq >= nq) ni[u] = none;
      }
#pragma unroll
      for (int u = 0; u < kBwdStage; ++u)
#pragma unroll
        for (int c = 0; c < CH; ++c) add4(acc + (size_t)c * T, n0, span, ci[u], cv[u][c]);
#pragma unroll
      for (int u = 0; u < kBwdStage; ++u) {
        ci[u] = ni[u];
#pragma unroll
        for (int c = 0; c < CH; ++c) cv[u][c] = nv[u][c];
      }
    }
  } else {
    for (int e = threadIdx.x; e < MK; e += 256) {
      const int i = ib[e];
      if ((unsigned)(i - n0) < span)
        for (int c = 0; c < CH; ++c) atomicAdd(&acc[(size_t)c * T + i - n0], (double)g[(size_t)c * MK + e]);
    }
  }
  __syncthreads();
  for (int c = 0; c < CH; ++c) {
    float *dst = grad_points + (size_t)(bc + c) * N + n0;
    for (int i = threadIdx.x; i < (int)span; i += 256) dst[i] = (float)acc[(size_t)c * T + i];
  }
}

// -------------------------------------------------- fused relative-position + feature gather
// rel[b,a,j,k] = (s[b,idx,a] - q[b,j,a]) * inv  (a<3), one thread per (j,k), xyz gathered
// straight from the AoS support array (12 B per neighbour; the reference first transposes the
// cloud to [B,3,N] and then runs its generic gather three times over it).
__global__ __launch_bounds__(256) void group_rel_kernel(const float *__restrict__ query_xyz,
                                                        const float *__restrict__ support_xyz,
                                                        const int *__restrict__ idx, int N, int M,
                                                        int K, float inv, int normalize,
                                                        float *__restrict__ rel) {
  const int b = blockIdx.y;
  const int MK = M * K;
  const float *q = query_xyz + (size_t)b * M * 3;
  const float *s = support_xyz + (size_t)b * N * 3;
  const int *ib = idx + (size_t)b * MK;
  float *r = rel + (size_t)b * 3 * MK;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < MK; e += gridDim.x * 256) {
    const int j = e / K;
    const int i = ib[e];
    float dx = s[i * 3 + 0] - q[j * 3 + 0];
    float dy = s[i * 3 + 1] - q[j * 3 + 1];
    float dz = s[i * 3 + 2] - q[j * 3 + 2];
    if (normalize) {
      dx *= inv;
      dy *= inv;
      dz *= inv;
    }
    r[e] = dx;
    r[MK + e] = dy;
    r[2 * MK + e] = dz;
  }
}

// Same output, for clouds whose coordinates fit LDS (N <= 5461): the support cloud is staged once per block
// (it is gathered 3*MK times), a thread owns four consecutive slots -- one int4 index load, three float4 stores.
constexpr int kRelSlotsPerBlock = 8192;
__global__ __launch_bounds__(256) void group_rel_lds_kernel(const float *__restrict__ query_xyz,
                                                            const float *__restrict__ support_xyz,
                                                            const int *__restrict__ idx, int N, int M, int K,
                                                            float inv, int normalize, float *__restrict__ rel) {
  extern __shared__ float sxyz[];  // [N*3]
  const int b = blockIdx.y;
  const int MK = M * K;
  const float *q = query_xyz + (size_t)b * M * 3;
  const float *s = support_xyz + (size_t)b * N * 3;
  for (int t = threadIdx.x; t < N * 3; t += 256) sxyz[t] = s[t];
  __syncthreads();
  const int4 *ib4 = reinterpret_cast<const int4 *>(idx + (size_t)b * MK);
  float *r = rel + (size_t)b * 3 * MK;
  const int e_begin = blockIdx.x * kRelSlotsPerBlock;
  const int e_end = e_begin + kRelSlotsPerBlock < MK ? e_begin + kRelSlotsPerBlock : MK;
  for (int e = e_begin + 4 * (int)threadIdx.x; e < e_end; e += 1024) {
    const int4 ii = ib4[e >> 2];
    const int iv[4] = {ii.x, ii.y, ii.z, ii.w};
    float o[3][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = (e + u) / K;
      const int i = (unsigned)iv[u] < (unsigned)N ? iv[u] : 0;  // an index outside the cloud (failed nearest query) reads point 0
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        float d = sxyz[i * 3 + a] - q[j * 3 + a];
        if (normalize) d *= inv;
        o[a][u] = d;
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
      store_stream(reinterpret_cast<float4 *>(r + (size_t)a * MK + e), make_float4(o[a][0], o[a][1], o[a][2], o[a][3]));
  }
}

static int launch_group_fwd(const float *points, const int *idx, int B, int C, int N, int M, int K,
                            float *out, hipStream_t st) {
  const long long MKll = (long long)M * K;
  if (MKll > 0x7fffffffLL) return fail(CL3D_E_UNSUPPORTED, "group_points: M*K too large");
  const int MK = (int)MKll;
  if (N <= kMaxLdsRow) {
    // split the M*K row so that the grid has >= ~2048 blocks, but keep a slice >= 4 rows' worth
    // of output so the staging read stays a small fraction of the traffic
    int want = ceil_div(2048, B * C);
    int max_split = MK / (4 * (N > 1024 ? N : 1024));
    if (max_split < 1) max_split = 1;
    int split = want < max_split ? want : max_split;
    if (split < 1) split = 1;
    int chunk = ceil_div(MK, split);
    chunk = (chunk + 1023) & ~1023;  // whole float4 sweeps of the block
    split = ceil_div(MK, chunk);
    dim3 grid(split, B * C);
    hipLaunchKernelGGL(group_fwd_lds_kernel, grid, dim3(256), (size_t)N * sizeof(float), st, points,
                       idx, C, N, MK, chunk, out);
  } else {
    int gx = ceil_div(MK, 256 * 8);
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    dim3 grid(gx, B * C);
    hipLaunchKernelGGL(group_fwd_direct_kernel, grid, dim3(256), 0, st, points, idx, C, N, MK, out);
  }
  return check_launch("cl3d_group_points");
}

}  // namespace cl3d

extern "C" int cl3d_group_points(const float *points, const int32_t *idx, int B, int C, int N,
                                 int M, int K, float *out, cl3d_stream_t stream) {
  CL3D_REQUIRE(B >= 0 && C >= 0 && N >= 1 && M >= 0 && K >= 0, "group_points: bad sizes");
  if (B == 0 || C == 0 || M == 0 || K == 0) return CL3D_OK;
  CL3D_REQUIRE(points && idx && out, "group_points: null pointer");
  CL3D_REQUIRE((long long)B * C <= 65535, "group_points: B*C=%lld exceeds grid.y limit", (long long)B * C);
  return cl3d::launch_group_fwd(points, idx, B, C, N, M, K, out, (hipStream_t)stream);
}

extern "C" int cl3d_group_points_grad(const float *grad_out, const int32_t *idx, int B, int C,
                                      int N, int M, int K, float *grad_points, void *ws,
                                      size_t ws_bytes, cl3d_stream_t stream) {
  (void)ws;
  (void)ws_bytes;
  CL3D_REQUIRE(B >= 0 && C >= 0 && N >= 1 && M >= 0 && K >= 0, "group_points_grad: bad sizes");
  if (B == 0 || C == 0) return CL3D_OK;
  CL3D_REQUIRE(grad_points, "group_points_grad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const long long MKll = (long long)M * K;
  if (MKll > 0x7fffffffLL) return cl3d::fail(CL3D_E_UNSUPPORTED, "group_points_grad: M*K too large");
  const int MK = (int)MKll;
  if (MK == 0) {
    hipError_t e = hipMemsetAsync(grad_points, 0, (size_t)B * C * N * sizeof(float), st);
    if (e != hipSuccess) return cl3d::fail(CL3D_E_LAUNCH, "group_points_grad: memset: %s", hipGetErrorString(e));
    return CL3D_OK;
  }
  CL3D_REQUIRE(grad_out && idx, "group_points_grad: null pointer");
  // Each block owns one (cloud, channel) row (two when they fit) and one tile of T support indices, accumulated in ONE
  // shared LDS row of doubles with ds_add_f64 (see group_bwd_lds_kernel: the sum is exact in double for the handful of
  // terms a row element gets, so the float result does not depend on arrival order except on a rounding boundary).
  // N <= 8192: a single tile; larger clouds sweep the grad_out row once per tile.  No global atomics.
  CL3D_REQUIRE((long long)B * C <= 0x7fffffffLL, "group_points_grad: B*C too large");
  const int kTile = 8192;  // doubles per block: 64 KiB, the no-opt-in dynamic LDS limit
  const int T = N <= kTile ? N : kTile;
  const int ntiles = cl3d::ceil_div(N, T);
  CL3D_REQUIRE(ntiles <= 65535, "group_points_grad: N too large");
  if ((C & 1) == 0 && T <= 4096)  // two channel rows per block share the index stream (2 x 32 KiB of LDS)
    hipLaunchKernelGGL(cl3d::group_bwd_lds_kernel<2>, dim3(B * C / 2, ntiles), dim3(256), (size_t)2 * T * sizeof(double), st,
                       grad_out, idx, C, N, MK, T, grad_points);
  else
    hipLaunchKernelGGL(cl3d::group_bwd_lds_kernel<1>, dim3(B * C, ntiles), dim3(256), (size_t)T * sizeof(double), st,
                       grad_out, idx, C, N, MK, T, grad_points);
  return cl3d::check_launch("cl3d_group_points_grad");
}

extern "C" int cl3d_group_xyz_features(const float *query_xyz, const float *support_xyz,
                                       const float *features, const int32_t *idx, int B, int C,
                                       int N, int M, int K, float radius, int normalize_xyz,
                                       float *rel, float *grouped, cl3d_stream_t stream) {
  CL3D_REQUIRE(B >= 0 && N >= 1 && M >= 0 && K >= 0 && C >= 0, "group_xyz_features: bad sizes");
  if (B == 0 || M == 0 || K == 0) return CL3D_OK;
  CL3D_REQUIRE(query_xyz && support_xyz && idx && rel, "group_xyz_features: null pointer");
  CL3D_REQUIRE(B <= 65535, "group_xyz_features: B exceeds grid.y limit");
  hipStream_t st = (hipStream_t)stream;
  const long long MKll = (long long)M * K;
  if (MKll > 0x7fffffffLL / 3) return cl3d::fail(CL3D_E_UNSUPPORTED, "group_xyz_features: M*K too large");
  int gx = cl3d::ceil_div((int)MKll, 256 * 4);
  gx = gx < 1 ? 1 : (gx > 4096 ? 4096 : gx);
  // the reference computes grouped_xyz /= radius through ATen's scalar-divide, which on the GPU
  // multiplies by the float reciprocal (BinaryDivTrueKernel) -- same here.
  const float inv = 1.0f / radius;
  if ((size_t)N * 12 <= 64 * 1024 && (MKll & 3) == 0) {
    hipLaunchKernelGGL(cl3d::group_rel_lds_kernel, dim3(cl3d::ceil_div((int)MKll, cl3d::kRelSlotsPerBlock), B), dim3(256),
                       (size_t)N * 12, st, query_xyz, support_xyz, idx, N, M, K, inv, normalize_xyz, rel);
  } else {
    hipLaunchKernelGGL(cl3d::group_rel_kernel, dim3(gx, B), dim3(256), 0, st, query_xyz, support_xyz, idx, N, M, K, inv, normalize_xyz, rel);
  }
  int rc = cl3d::check_launch("cl3d_group_xyz_features(rel)");
  if (rc != CL3D_OK) return rc;
  if (features != nullptr && C > 0) {
    CL3D_REQUIRE(grouped, "group_xyz_features: grouped is null");
    CL3D_REQUIRE((long long)B * C <= 65535, "group_xyz_features: B*C exceeds grid.y limit");
    return cl3d::launch_group_fwd(features, idx, B, C, N, M, K, grouped, st);
  }
  return CL3D_OK;
}

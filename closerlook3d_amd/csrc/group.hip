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
//   * backward: the mirror image.  One block owns one (cloud, channel) gradient row; each of its
//     W waves streams a fixed slice of grad_out (float4, coalesced) and accumulates into its own
//     private LDS row (tag-resolved read-add-write, see below); the W rows are combined in a fixed
//     order.  No atomics anywhere, and the summation order does not depend on scheduling.
//   * rows that do not fit LDS (N > kMaxLdsRow = 16384): the forward takes the direct
//     global-memory gather below; the backward tiles the support range, one LDS tile per block.
#include "cl3d_common.h"

namespace cl3d {

constexpr int kMaxLdsRow = 16384;  // floats: 64 KiB, the no-opt-in dynamic LDS limit
constexpr int kUnroll = 4;

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
        o4[g + u * 256] = v;
      }
    }
    for (; g < g1; g += 256) {
      const int4 ii = i4[g];
      float4 v;
      v.x = row[ii.x];
      v.y = row[ii.y];
      v.z = row[ii.z];
      v.w = row[ii.w];
      o4[g] = v;
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
// LDS float atomics are not an option on gfx950: ds_add_f32 sustains 0.33 op/clk/CU under random
// addresses (measured, scripts/microbench_lds.hip; ds_add_u32: 8.1, plain read-add-write: 4.3).  So the
// scatter is a plain read-add-write into a row PRIVATE to the wave, with same-instruction address
// collisions resolved by a tag word per accumulator: every lane writes a unique tag, reads it back, and
// only the lane whose tag survived performs its add; the others go round again (2.8 op/clk/CU
// measured, and the order in which colliding lanes add is the hardware's fixed lane order).
struct AccSlot {
  float sum;
  unsigned tag;
};
// LDS-address-space volatile views: `volatile` because the tag must really be re-read from LDS after
// all lanes have written theirs (the compiler would otherwise forward a lane's own store to its load)
// and the accesses must stay in program order; the explicit address space keeps them ds_* instructions
// (a plain volatile pointer degrades to flat_load/flat_store).
typedef __attribute__((address_space(3))) volatile float lds_vf32;
typedef __attribute__((address_space(3))) volatile unsigned lds_vu32;

__device__ __forceinline__ void scatter_add4(AccSlot *mine, int n0, unsigned span, const int4 &ii,
                                             const float4 &v, unsigned tag) {
  const int id[4] = {ii.x - n0, ii.y - n0, ii.z - n0, ii.w - n0};
  const float val[4] = {v.x, v.y, v.z, v.w};
  bool todo[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) todo[c] = (unsigned)id[c] < span;
  while (__ballot(todo[0] | todo[1] | todo[2] | todo[3])) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (todo[c]) *(lds_vu32 *)(&mine[id[c]].tag) = tag + c;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (todo[c] && *(lds_vu32 *)(&mine[id[c]].tag) == tag + c) {
        lds_vf32 *ps = (lds_vf32 *)(&mine[id[c]].sum);
        *ps = *ps + val[c];
        todo[c] = false;
      }
    }
  }
}

template <int W>
__global__ __launch_bounds__(64 * W) void group_bwd_lds_kernel(const float *__restrict__ grad_out,
                                                              const int *__restrict__ idx, int C,
                                                              int N, int MK, int T,
                                                              float *__restrict__ grad_points) {
  extern __shared__ AccSlot acc[];  // [W][T]: this block owns support indices [n0, n0+T)
  const int bc = blockIdx.x;
  const int b = bc / C;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int n0 = blockIdx.y * T;
  const unsigned span = (unsigned)(N - n0 < T ? N - n0 : T);
  for (int i = threadIdx.x; i < W * T; i += 64 * W) {
    acc[i].sum = 0.0f;
    acc[i].tag = 0xffffffffu;
  }
  __syncthreads();

  AccSlot *mine = acc + (size_t)wave * T;
  const float *g = grad_out + (size_t)bc * MK;
  const int *ib = idx + (size_t)b * MK;
  // fixed slice per wave, multiple of 4 elements
  int per = ((MK + W - 1) / W + 3) & ~3;
  const int e0 = wave * per;
  int e1 = e0 + per;
  e1 = e1 < MK ? e1 : MK;
  unsigned tag = (unsigned)lane << 2;  // unique per (lane, component); bumped by 256 per use
  if ((MK & 3) == 0) {
    const int4 *i4 = reinterpret_cast<const int4 *>(ib);
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    const int q1 = e1 >> 2;
    int q = (e0 >> 2) + lane;
    // 2*kUnroll independent 16-byte loads in flight per lane (the stream comes from HBM)
    for (; q + (kUnroll - 1) * 64 < q1; q += kUnroll * 64) {
      int4 ii[kUnroll];
      float4 vv[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        ii[u] = i4[q + u * 64];
        vv[u] = g4[q + u * 64];
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        scatter_add4(mine, n0, span, ii[u], vv[u], tag);
        tag += 256;
      }
    }
    for (; q < q1; q += 64) {
      scatter_add4(mine, n0, span, i4[q], g4[q], tag);
      tag += 256;
    }
  } else {
    for (int e = e0 + lane; e < e1; e += 64) {
      const int4 ii = make_int4(ib[e], -1, -1, -1);
      scatter_add4(mine, n0, span, ii, make_float4(g[e], 0.f, 0.f, 0.f), tag);
      tag += 256;
    }
  }
  __syncthreads();
  float *dst = grad_points + (size_t)bc * N + n0;
  for (int i = threadIdx.x; i < (int)span; i += 64 * W) {
    float s = acc[i].sum;
#pragma unroll
    for (int w = 1; w < W; ++w) s += acc[(size_t)w * T + i].sum;
    dst[i] = s;
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
  // Each block owns one (cloud, channel) row and one tile of T support indices, accumulated in W
  // private LDS rows (one per wave, combined in a fixed order).  N <= 16384: a single tile; larger
  // clouds sweep the grad_out row once per tile (still no global atomics, still order-fixed).
  CL3D_REQUIRE((long long)B * C <= 0x7fffffffLL, "group_points_grad: B*C too large");
  const int kTile = 8192;  // accumulators per wave: 64 KiB of {sum, tag} pairs
  const int T = N <= kTile ? N : kTile;
  const int ntiles = cl3d::ceil_div(N, T);
  CL3D_REQUIRE(ntiles <= 65535, "group_points_grad: N too large");
  const size_t row = (size_t)T * sizeof(cl3d::AccSlot);
  const dim3 grid(B * C, ntiles);
  // one wave per block unless the row is small: occupancy comes from many resident blocks, and every
  // wave keeps 2*kUnroll 16-byte loads in flight
  if (row * 2 <= 32 * 1024 && MK >= 2048)
    hipLaunchKernelGGL(cl3d::group_bwd_lds_kernel<2>, grid, dim3(128), row * 2, st, grad_out, idx, C, N, MK, T, grad_points);
  else
    hipLaunchKernelGGL(cl3d::group_bwd_lds_kernel<1>, grid, dim3(64), row, st, grad_out, idx, C, N, MK, T, grad_points);
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
  hipLaunchKernelGGL(cl3d::group_rel_kernel, dim3(gx, B), dim3(256), 0, st, query_xyz, support_xyz, idx, N, M, K, inv, normalize_xyz, rel);
  int rc = cl3d::check_launch("cl3d_group_xyz_features(rel)");
  if (rc != CL3D_OK) return rc;
  if (features != nullptr && C > 0) {
    CL3D_REQUIRE(grouped, "group_xyz_features: grouped is null");
    CL3D_REQUIRE((long long)B * C <= 65535, "group_xyz_features: B*C exceeds grid.y limit");
    return cl3d::launch_group_fwd(features, idx, B, C, N, M, K, grouped, st);
  }
  return CL3D_OK;
}

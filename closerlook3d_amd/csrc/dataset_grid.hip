// dataset_grid.hip -- the dataset-side grid subsampling on the GPU: voxel barycentres with
// feature means and majority labels, what the reference's loaders call per cloud / per room on the host
// (datasets/data_utils.py:12-30 -> ops/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106).
//
// Same arithmetic as the reference, operation by operation: voxel = floor((p - origin)/dl) per axis in float
// (IEEE divide), origin = floor(min * (1/dl)) * dl; per voxel, IN ORIGINAL POINT ORDER, float sums of coordinates
// and features; barycentre = sum * (float)(1.0/count), feature mean = sum / (float)count; label = the most
// frequent one per label column.  The reference walks an unordered_map, so its output order -- and which of
// several equally frequent labels wins -- is implementation-defined; here voxels come out in ascending
// (iz, iy, ix) order and ties go to the smallest label (as in oracle/cl3d_oracle.c).
//
// Pipeline (one cloud of N points, no host round trip): bbox partials -> origin and grid dims -> 64-bit voxel
// keys -> stable radix sort of (key, index) (rocPRIM; stability = original point order inside a voxel) ->
// voxel heads + inclusive scan = output row of every voxel -> one thread per voxel folds its members in order.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "ball_query.h"

namespace cl3d {

struct DgState {
  float ox, oy, oz, pad;
  long long nx, ny;
};

__global__ __launch_bounds__(256) void dg_bbox_partial_kernel(const float *__restrict__ p, int n, float *__restrict__ partial) {
  __shared__ float red[6][4];
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = p[3 * (size_t)i + a];
      mn[a] = v < mn[a] ? v : mn[a];
      mx[a] = v > mx[a] ? v : mx[a];
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float a0 = __shfl_xor(mn[a], o, 64), a1 = __shfl_xor(mx[a], o, 64);
      mn[a] = a0 < mn[a] ? a0 : mn[a];
      mx[a] = a1 > mx[a] ? a1 : mx[a];
    }
    if ((threadIdx.x & 63) == 0) {
      red[a][threadIdx.x >> 6] = mn[a];
      red[3 + a][threadIdx.x >> 6] = mx[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[threadIdx.x][0];
    for (int w = 1; w < 4; ++w) {
      const float o = red[threadIdx.x][w];
      v = threadIdx.x < 3 ? (o < v ? o : v) : (o > v ? o : v);
    }
    partial[blockIdx.x * 6 + threadIdx.x] = v;
  }
}

__global__ void dg_bbox_final_kernel(const float *__restrict__ partial, int g, float dl, DgState *st) {
  if (threadIdx.x != 0) return;
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int b = 0; b < g; ++b)
    for (int a = 0; a < 3; ++a) {
      mn[a] = partial[b * 6 + a] < mn[a] ? partial[b * 6 + a] : mn[a];
      mx[a] = partial[b * 6 + 3 + a] > mx[a] ? partial[b * 6 + 3 + a] : mx[a];
    }
  const float inv = 1 / dl;
  st->ox = floorf(mn[0] * inv) * dl;
  st->oy = floorf(mn[1] * inv) * dl;
  st->oz = floorf(mn[2] * inv) * dl;
  st->nx = (long long)floorf((mx[0] - st->ox) / dl) + 1;
  st->ny = (long long)floorf((mx[1] - st->oy) / dl) + 1;
}

__global__ __launch_bounds__(256) void dg_keys_kernel(const float *__restrict__ p, int n, float dl, const DgState *st,
                                                      unsigned long long *__restrict__ keys, unsigned *__restrict__ vals) {
  const float ox = st->ox, oy = st->oy, oz = st->oz;
  const long long nx = st->nx, ny = st->ny;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const long long ix = (long long)floorf((p[3 * (size_t)i] - ox) / dl);
    const long long iy = (long long)floorf((p[3 * (size_t)i + 1] - oy) / dl);
    const long long iz = (long long)floorf((p[3 * (size_t)i + 2] - oz) / dl);
    keys[i] = (unsigned long long)(ix + nx * iy + nx * ny * iz);
    vals[i] = (unsigned)i;
  }
}

__global__ __launch_bounds__(256) void dg_heads_kernel(const unsigned long long *__restrict__ keys, int n, int *__restrict__ flag) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void dg_fold_kernel(const float *__restrict__ p, const float *__restrict__ feat,
                                                      const int *__restrict__ lab, int n, int fdim, int ldim,
                                                      const unsigned long long *__restrict__ keys,
                                                      const unsigned *__restrict__ order, const int *__restrict__ flag,
                                                      const int *__restrict__ rank, float *__restrict__ sub_p,
                                                      float *__restrict__ sub_f, int *__restrict__ sub_l, int *__restrict__ count) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    if (i == n - 1) *count = rank[i];
    if (flag[i] == 0) continue;
    const int r = rank[i] - 1;
    const unsigned long long key = keys[i];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    float *fo = sub_f + (size_t)r * fdim;
    for (int f = 0; f < fdim; ++f) fo[f] = 0.f;
    int e = i;
    for (; e < n && keys[e] == key; ++e) {  // members in original point order (stable sort)
      const size_t j = order[e];
      sx += p[3 * j];
      sy += p[3 * j + 1];
      sz += p[3 * j + 2];
      for (int f = 0; f < fdim; ++f) fo[f] += feat[j * fdim + f];
    }
    const int cnt = e - i;
    const float rc = (float)(1.0 / cnt);
    sub_p[3 * (size_t)r] = sx * rc;
    sub_p[3 * (size_t)r + 1] = sy * rc;
    sub_p[3 * (size_t)r + 2] = sz * rc;
    for (int f = 0; f < fdim; ++f) fo[f] /= (float)cnt;
    for (int c = 0; c < ldim; ++c) {  // majority label, smallest on ties: O(cnt^2), voxels hold tens of points
      int best = 0, best_cnt = 0;
      for (int t = i; t < e; ++t) {
        const int lb = lab[(size_t)order[t] * ldim + c];
        int k = 0;
        for (int u = i; u < e; ++u) k += lab[(size_t)order[u] * ldim + c] == lb;
        if (k > best_cnt || (k == best_cnt && lb < best)) {
          best = lb;
          best_cnt = k;
        }
      }
      sub_l[(size_t)r * ldim + c] = best;
    }
  }
}

struct DgLayout {
  size_t keys_in, keys_out, vals_in, vals_out, flag, rank, partial, state, temp, temp_bytes, total;
};

static DgLayout dg_layout(int n) {
  DgLayout l{};
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += (bytes + 255) & ~(size_t)255;
    return at;
  };
  const size_t N = (size_t)(n > 0 ? n : 1);
  l.keys_in = take(N * 8); l.keys_out = take(N * 8); l.vals_in = take(N * 4); l.vals_out = take(N * 4);
  l.flag = take(N * 4); l.rank = take(N * 4); l.partial = take(1024 * 6 * 4); l.state = take(sizeof(DgState));
  size_t sort_bytes = 0, scan_bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                  (const unsigned *)nullptr, (unsigned *)nullptr, N, 0, 64, (hipStream_t)0);
  (void)rocprim::inclusive_scan(nullptr, scan_bytes, (const int *)nullptr, (int *)nullptr, N, rocprim::plus<int>(), (hipStream_t)0);
  l.temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  l.temp = take(l.temp_bytes);
  l.total = off;
  return l;
}

size_t dataset_grid_workspace(int n) { return dg_layout(n).total; }

}  // namespace cl3d

extern "C" int cl3d_dataset_grid_subsampling(const float *points, const float *features, const int32_t *labels, int N,
                                             int fdim, int ldim, float sampleDl, float *sub_points, float *sub_features,
                                             int32_t *sub_labels, int32_t *count, void *ws, size_t ws_bytes,
                                             cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(N >= 0 && fdim >= 0 && ldim >= 0 && sampleDl > 0.f, "dataset_grid_subsampling: bad arguments");
  CL3D_REQUIRE(count, "dataset_grid_subsampling: null count");
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) {
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int32_t), st);
    return e == hipSuccess ? CL3D_OK : fail(CL3D_E_LAUNCH, "dataset_grid_subsampling: memset: %s", hipGetErrorString(e));
  }
  CL3D_REQUIRE(points && sub_points && (fdim == 0 || (features && sub_features)) && (ldim == 0 || (labels && sub_labels)),
               "dataset_grid_subsampling: null pointer");
  const DgLayout l = dg_layout(N);
  if (!ws || ws_bytes < l.total) return fail(CL3D_E_WORKSPACE, "dataset_grid_subsampling: workspace %zu < %zu", ws_bytes, l.total);
  char *base = static_cast<char *>(ws);
  auto *keys_in = reinterpret_cast<unsigned long long *>(base + l.keys_in);
  auto *keys_out = reinterpret_cast<unsigned long long *>(base + l.keys_out);
  auto *vals_in = reinterpret_cast<unsigned *>(base + l.vals_in);
  auto *vals_out = reinterpret_cast<unsigned *>(base + l.vals_out);
  int *flag = reinterpret_cast<int *>(base + l.flag);
  int *rank = reinterpret_cast<int *>(base + l.rank);
  float *partial = reinterpret_cast<float *>(base + l.partial);
  DgState *state = reinterpret_cast<DgState *>(base + l.state);
  int g = ceil_div(N, 256 * 8);
  g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
  hipLaunchKernelGGL(dg_bbox_partial_kernel, dim3(g), dim3(256), 0, st, points, N, partial);
  hipLaunchKernelGGL(dg_bbox_final_kernel, dim3(1), dim3(64), 0, st, partial, g, sampleDl, state);
  int gx = ceil_div(N, 256 * 4);
  gx = gx < 1 ? 1 : (gx > 8192 ? 8192 : gx);
  hipLaunchKernelGGL(dg_keys_kernel, dim3(gx), dim3(256), 0, st, points, N, sampleDl, state, keys_in, vals_in);
  size_t tb = l.temp_bytes;
  hipError_t e = rocprim::radix_sort_pairs(base + l.temp, tb, (const unsigned long long *)keys_in, keys_out,
                                           (const unsigned *)vals_in, vals_out, (size_t)N, 0, 64, st);
  if (e != hipSuccess) return fail(CL3D_E_LAUNCH, "dataset_grid_subsampling: radix sort: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(dg_heads_kernel, dim3(gx), dim3(256), 0, st, keys_out, N, flag);
  tb = l.temp_bytes;
  e = rocprim::inclusive_scan(base + l.temp, tb, (const int *)flag, rank, (size_t)N, rocprim::plus<int>(), st);
  if (e != hipSuccess) return fail(CL3D_E_LAUNCH, "dataset_grid_subsampling: scan: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(dg_fold_kernel, dim3(gx), dim3(256), 0, st, points, features, labels, N, fdim, ldim, keys_out,
                     vals_out, flag, rank, sub_points, sub_features, sub_labels, count);
  return check_launch("cl3d_dataset_grid_subsampling");
}

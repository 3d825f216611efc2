// Micro-benchmark: a lane's random point coordinates as three dword loads out of [N,3] floats against one dwordx4 load
// out of a [N,4] table (2.1 M random points of 16 clouds x 4096, like the staging of the fused gather passes).
//   hipcc --offload-arch=gfx950 -O3 -o gather_xyz gather_xyz.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE, int U>
__global__ __launch_bounds__(256) void xyz_kernel(const float *__restrict__ xyz3, const float4 *__restrict__ xyz4,
                                                  const int *__restrict__ index, int per_cloud, long long total,
                                                  float *__restrict__ out) {
  const long long t0 = ((long long)blockIdx.x * 256 + threadIdx.x) * U;
  float acc = 0.f;
  int id[U];
#pragma unroll
  for (int u = 0; u < U; ++u) id[u] = t0 + u < total ? index[t0 + u] : 0;
  if (MODE == 0) {
    float x[U], y[U], z[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      x[u] = xyz3[(size_t)id[u] * 3 + 0];
      y[u] = xyz3[(size_t)id[u] * 3 + 1];
      z[u] = xyz3[(size_t)id[u] * 3 + 2];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += x[u] + y[u] + z[u];
  } else {
    float4 p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) p[u] = xyz4[id[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += p[u].x + p[u].y + p[u].z;
  }
  out[(long long)blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  const int clouds = 16, rows = 4096, K = 32;
  const long long total = (long long)clouds * rows * K;
  std::vector<int> h(total);
  srand(1);
  for (long long t = 0; t < total; ++t) h[t] = (int)(t / ((long long)rows * K)) * rows + rand() % rows;  // inside the slot's cloud
  int *index;
  float *xyz3, *out;
  float4 *xyz4;
  hipMalloc(&index, total * 4);
  hipMalloc(&xyz3, (size_t)clouds * rows * 12);
  hipMalloc(&xyz4, (size_t)clouds * rows * 16);
  hipMalloc(&out, total * 4);
  hipMemset(xyz3, 0, (size_t)clouds * rows * 12);
  hipMemset(xyz4, 0, (size_t)clouds * rows * 16);
  hipMemcpy(index, h.data(), total * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int mode = 0; mode < 2; ++mode)
    for (int u = 1; u <= 4; u *= 4) {
      float best = 1e9f;
      const int grid = (int)((total / u + 255) / 256);
      for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(a);
        if (mode == 0 && u == 1) hipLaunchKernelGGL((xyz_kernel<0, 1>), dim3(grid), dim3(256), 0, 0, xyz3, xyz4, index, rows, total, out);
        if (mode == 0 && u == 4) hipLaunchKernelGGL((xyz_kernel<0, 4>), dim3(grid), dim3(256), 0, 0, xyz3, xyz4, index, rows, total, out);
        if (mode == 1 && u == 1) hipLaunchKernelGGL((xyz_kernel<1, 1>), dim3(grid), dim3(256), 0, 0, xyz3, xyz4, index, rows, total, out);
        if (mode == 1 && u == 4) hipLaunchKernelGGL((xyz_kernel<1, 4>), dim3(grid), dim3(256), 0, 0, xyz3, xyz4, index, rows, total, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
      }
      printf("%s, %d per thread: %7.1f us for %lld points\n", mode == 0 ? "3 x dword  out of [N,3]" : "1 x dwordx4 out of [N,4]", u, best * 1e3, total);
    }
  return 0;
}

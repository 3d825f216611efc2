// Micro-benchmark: 256-byte row gathers (16 lanes x 16 B) out of a point-major table, as the PointWiseMLP gather passes
// issue them, for different row pitches / offsets -- does reading only one 256-byte half of 512-byte rows (the G or the H
// half of ght [B,N,2Co]) use the L2 unevenly?   hipcc --offload-arch=gfx950 -O3 -o gather_pitch gather_pitch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int KB>
__global__ __launch_bounds__(256) void gather_kernel(const char *__restrict__ table, const int *__restrict__ index,
                                                     int rows_per_cloud, int per_group, unsigned pitch, unsigned offset,
                                                     float4 *__restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, cl = lane & 15;
  const long long group = ((long long)blockIdx.x * 4 + wave) * 4 + g;
  const int *my = index + group * per_group;
  const int cloud = blockIdx.x % 16;  // XCD-like placement: consecutive workgroups on different clouds
  const char *base = table + (size_t)cloud * rows_per_cloud * pitch + offset + cl * 16;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int e = 0; e < per_group; e += KB) {
    int id[KB];
    float4 v[KB];
#pragma unroll
    for (int u = 0; u < KB; ++u) id[u] = my[e + u];
#pragma unroll
    for (int u = 0; u < KB; ++u) v[u] = *reinterpret_cast<const float4 *>(base + (unsigned)id[u] * pitch);
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
    }
  }
  out[group * 16 + cl] = acc;
}

int main() {
  const int clouds = 16, rows = 4096, per_group = 32;
  const long long groups = (long long)clouds * rows;  // one group per (cloud, point): 2.1 M row gathers
  std::vector<int> h((size_t)groups * per_group);
  srand(1);
  for (auto &x : h) x = rand() % rows;
  int *index;
  float4 *out;
  char *table;
  const size_t tbytes = (size_t)clouds * rows * 1024;
  hipMalloc(&index, h.size() * 4);
  hipMalloc(&out, groups * 16 * sizeof(float4));
  hipMalloc(&table, tbytes);
  hipMemset(table, 0, tbytes);
  hipMemcpy(index, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  struct Cfg { unsigned pitch, offset; const char *what; };
  const Cfg cfgs[] = {{256, 0, "pitch 256 (rows packed)"},      {512, 0, "pitch 512, first half (G of ght)"},
                      {512, 256, "pitch 512, second half (H)"}, {768, 0, "pitch 768"},
                      {1024, 0, "pitch 1024"},                  {640, 0, "pitch 640"}};
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int grid = (int)(groups / 16);
  for (int kb = 4; kb <= 8; kb += 4)
    for (const Cfg &c : cfgs) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(a);
        if (kb == 4) hipLaunchKernelGGL(gather_kernel<4>, dim3(grid), dim3(256), 0, 0, table, index, rows, per_group, c.pitch, c.offset, out);
        else hipLaunchKernelGGL(gather_kernel<8>, dim3(grid), dim3(256), 0, 0, table, index, rows, per_group, c.pitch, c.offset, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
      }
      printf("KB=%d  %-36s %7.1f us   %.1f TB/s of rows\n", kb, c.what, best * 1e3, groups * per_group * 256.0 / best / 1e9);
    }
  return 0;
}

// Microbenchmark: LDS scatter-accumulate throughput on gfx950 (guides the scatter-add design).
// hipcc --offload-arch=gfx950 -O3 scripts/microbench_lds.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(256) void k(const int* __restrict__ idx, int n_idx, int iters, float* out) {
  __shared__ float acc[4096];
  __shared__ unsigned uacc[4096];
  __shared__ double dacc[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) { acc[i] = 0; uacc[i] = 0; dacc[i] = 0; }
  __syncthreads();
  int base = (blockIdx.x * 256 + threadIdx.x) * 8;
  int ii[8];
  for (int u = 0; u < 8; ++u) ii[u] = idx[(base + u) % n_idx];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int a = (ii[u] + it * 97) & 4095;
      if (MODE == 0) atomicAdd(&acc[a], 1.0f);                 // ds_add_f32
      else if (MODE == 1) atomicAdd(&uacc[a], 1u);              // ds_add_u32
      else if (MODE == 2) acc[a] = acc[a] + 1.0f;               // plain RMW (racy; rate only)
      else if (MODE == 3) { float v = acc[a]; asm volatile("" :: "v"(v)); }  // read only
      else if (MODE == 4) atomicAdd(&dacc[a], 1.0);            // ds_add_f64
      else if (MODE == 5) {                                     // tag-resolved plain RMW (exact under intra-wave conflicts)
        const unsigned tag = ((unsigned)(it * 8 + u) << 8) | (threadIdx.x & 255u);
        bool todo = true;
        while (__ballot(todo)) {
          if (todo) uacc[a] = tag;
          if (todo && uacc[a] == tag) { acc[a] = acc[a] + 1.0f; todo = false; }
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = acc[0] + (float)uacc[0];
}

template <int MODE>
void run(const char* name, int* d_idx, int n_idx, float* d_out, int blocks) {
  const int iters = 2000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_idx, n_idx, 10, d_out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_idx, n_idx, iters, d_out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ops = (double)blocks * 256 * 8 * iters;
  printf("%-28s blocks=%5d  %8.3f ms  %8.1f Gop/s  %6.2f op/clk/CU (2.4GHz,256CU)\n", name, blocks, ms, ops / ms / 1e6,
         ops / (ms * 1e-3) / 2.4e9 / 256);
}

int main() {
  const int n_idx = 1 << 20;
  std::vector<int> h(n_idx);
  srand(1);
  for (auto& v : h) v = rand() & 4095;
  int* d_idx; float* d_out;
  hipMalloc(&d_idx, n_idx * 4); hipMalloc(&d_out, 1 << 20);
  hipMemcpy(d_idx, h.data(), n_idx * 4, hipMemcpyHostToDevice);
  for (int blocks : {256, 1024, 2048}) {
    run<0>("ds_add_f32 random", d_idx, n_idx, d_out, blocks);
    run<1>("ds_add_u32 random", d_idx, n_idx, d_out, blocks);
    run<2>("plain read+add+write random", d_idx, n_idx, d_out, blocks);
    run<3>("ds_read_b32 random", d_idx, n_idx, d_out, blocks);
    run<4>("ds_add_f64 random", d_idx, n_idx, d_out, blocks);
    run<5>("tag-resolved RMW random", d_idx, n_idx, d_out, blocks);
  }
  return 0;
}

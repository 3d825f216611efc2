// Issue rate of packed FP32 (v_pk_fma_f32 / v_pk_add_f32) against scalar v_fma_f32 on gfx950: the same number of
// floating-point operations on independent accumulators, 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off pk_rate.hip -o var/pk_rate && var/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, float a, float b) {
  float s[16];
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = (float)(threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = (f2){(float)(threadIdx.x + 2 * i), (float)(threadIdx.x + 2 * i + 1)};
  const f2 a2 = (f2)(a), b2 = (f2)(b);
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = __builtin_fmaf(s[i], a, b);
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], a2, b2);
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = s[i] + a;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = p[i] + a2;
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += s[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE> static float run(float *out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float *out;
  hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 20000;
  const char *names[4] = {"16 x v_fma_f32", "8 x v_pk_fma_f32", "16 x v_add_f32", "8 x v_pk_add_f32"};
  for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD: blocks of 4 waves, 256 CUs
    const int blocks = 256 * wps;
    float ms[4] = {run<0>(out, blocks, iters), run<1>(out, blocks, iters), run<2>(out, blocks, iters), run<3>(out, blocks, iters)};
    for (int m = 0; m < 4; ++m) {
      const double flops = (double)blocks * 256 * iters * 16 * (m < 2 ? 2 : 1);
      printf("waves/SIMD %d  %-18s %8.3f ms  %7.1f TFLOP/s  cycles per wave-instruction at 2.4 GHz: %.2f\n", wps, names[m], ms[m],
             flops / ms[m] * 1e-9, ms[m] * 1e-3 * 2.4e9 / ((double)iters * (m % 2 ? 8 : 16) * wps));
    }
  }
  return 0;
}

// Prototype: the PointWiseMLP's forward per-point product  ght[p][0..128) = sum_c F[c][p] * W[j][c]  (F channel-major
// [C=64][P], P = 65536 points, W [128][64]) with NO operand staging through LDS: the streamed operand's MFMA fragments
// are loaded straight from global memory (a 32-point block of one channel is 128 contiguous bytes = the lanes of one
// fragment), the weights sit in LDS for the workgroup's life, a wave owns a 32-point block x all 128 outputs and
// prefetches the next block's fragments while it multiplies.   hipcc --offload-arch=gfx950 -O3 -o skinny_gemm skinny_gemm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int C = 64, J = 128;

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void skinny_fwd(const float *__restrict__ F, const float *__restrict__ W,
                                                         float *__restrict__ out, int P, int nblocks) {
  __shared__ float wl[C * (J + 1)];  // W^T: wl[c][j], row stride J+1
  for (int t = threadIdx.x; t < C * J; t += 64 * WAVES) {
    const int j = t / C, c = t - j * C;
    wl[c * (J + 1) + j] = W[t];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int gw = blockIdx.x * WAVES + wave, nw = gridDim.x * WAVES;
  float a_cur[C / 2], a_nxt[C / 2];
  int blk = gw;
  if (blk < nblocks) {
#pragma unroll
    for (int s = 0; s < C / 2; ++s) a_cur[s] = F[(size_t)(2 * s + lh) * P + blk * 32 + lr];
  }
  for (; blk < nblocks; blk += nw) {
    const int nb = blk + nw;
    if (nb < nblocks) {
#pragma unroll
      for (int s = 0; s < C / 2; ++s) a_nxt[s] = F[(size_t)(2 * s + lh) * P + nb * 32 + lr];
    }
    f32x16 acc[4];
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[y][e] = 0.f;
#pragma unroll
    for (int s = 0; s < C / 2; ++s) {
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const float b = wl[(2 * s + lh) * (J + 1) + 32 * y + lr];
        acc[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s], b, acc[y], 0, 0, 0);
      }
    }
    // D: lane holds column j = 32y + lr, rows i = (e&3) + 8*(e>>2) + 4*lh of the 32-point block
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = (e & 3) + 8 * (e >> 2) + 4 * lh;
        out[(size_t)(blk * 32 + i) * J + 32 * y + lr] = acc[y][e];
      }
#pragma unroll
    for (int s = 0; s < C / 2; ++s) a_cur[s] = a_nxt[s];
  }
}

int main() {
  const int P = 65536, nblocks = P / 32;
  std::vector<float> hF((size_t)C * P), hW((size_t)J * C);
  srand(3);
  for (auto &x : hF) x = (rand() % 2001 - 1000) / 1000.f;
  for (auto &x : hW) x = (rand() % 2001 - 1000) / 1000.f;
  float *F, *W, *out;
  hipMalloc(&F, hF.size() * 4);
  hipMalloc(&W, hW.size() * 4);
  hipMalloc(&out, (size_t)P * J * 4);
  hipMemcpy(F, hF.data(), hF.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int cfg = 0; cfg < 6; ++cfg) {
    const int waves = cfg < 3 ? 4 : 2;
    const int grids[3] = {256, 512, 1024};
    const int grid = grids[cfg % 3];
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
      hipEventRecord(a);
      if (waves == 4) hipLaunchKernelGGL(skinny_fwd<4>, dim3(grid), dim3(256), 0, 0, F, W, out, P, nblocks);
      else hipLaunchKernelGGL(skinny_fwd<2>, dim3(grid), dim3(128), 0, 0, F, W, out, P, nblocks);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      if (rep > 0 && ms < best) best = ms;
    }
    printf("waves/workgroup %d, grid %4d: %6.1f us  (%.1f TFLOP/s, %.2f TB/s of 50.3 MB)\n", waves, grid, best * 1e3,
           2.0 * P * C * J / best / 1e9, 50.3e6 / best / 1e9);
  }
  std::vector<float> ho((size_t)P * J);
  hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int t = 0; t < 2000; ++t) {
    const int p = rand() % P, j = rand() % J;
    double r = 0;
    for (int c = 0; c < C; ++c) r += (double)hF[(size_t)c * P + p] * hW[(size_t)j * C + c];
    worst = fmax(worst, fabs(r - ho[(size_t)p * J + j]));
  }
  printf("max |error| on 2000 sampled outputs: %.3e\n", worst);
  return 0;
}

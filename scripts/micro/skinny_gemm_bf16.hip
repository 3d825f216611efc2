// Prototype (round 3, NOT yet run on a GPU -- the round's GPU budget was spent when it was written; compiles for gfx950,
// 66 registers, no spills): the bf16 form of the LDS-free forward product of csrc/mfma_gemm.hip (pwmlp_rows_nolds_kernel), for the
// step with the contraction in bf16, whose forward product is still the LDS-staged kernel and therefore waits behind
// the ball query.   ght[p][j] = sum_c bf16(F[c][p]) * bf16(W[j][c]),  F channel-major [C][P], W [J][C].
// A wave owns one 32-column tile of the output with its weight fragments resident in registers (C / 16 packs of 8 bf16)
// and walks 32-point blocks; a lane's A fragment for one v_mfma_f32_32x32x16_bf16 step is 8 channels of ONE point:
// eight 4-byte loads (each a 128-byte run across the 32 lanes of a half-wave), rounded to bf16 (RNE) and packed.  Same
// number of loads per block as the f32 kernel (C / 2 per lane), an eighth of its MFMA cycles.
//   hipcc --offload-arch=gfx950 -O3 -o skinny_gemm_bf16 skinny_gemm_bf16.hip && ./skinny_gemm_bf16
// Prints the worst error against a float64 sum over the same bf16-rounded inputs and the time per launch; the engine's
// staged bf16 forward product takes 22.8 us at this shape, the f32 LDS-free kernel 25 us.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CH16>  // C / 16 MFMA steps
__global__ __launch_bounds__(256, 5) void skinny_fwd_bf16(const float *__restrict__ F, const float *__restrict__ W,
                                                          float *__restrict__ out, int C, int P, int J, int nblocks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int j = 32 * (blockIdx.y * 4 + wave) + lr;
  if (32 * (blockIdx.y * 4 + wave) >= J) return;
  bf16x8 bw[CH16];
#pragma unroll
  for (int s = 0; s < CH16; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * s + 8 * lh + e;
      bw[s][e] = (__bf16)((j < J && k < C) ? W[(size_t)j * C + k] : 0.f);
    }
  const unsigned a_off = ((unsigned)(8 * lh) * (unsigned)P + (unsigned)lr) * 4u;  // lane part of every fragment address
  const unsigned o_off = ((unsigned)(4 * lh) * (unsigned)J + (unsigned)(j < J ? j : 0)) * 4u;
  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int n0 = __builtin_amdgcn_readfirstlane(blk * 32);
    const char *fb = reinterpret_cast<const char *>(F + n0);
    bf16x8 a[CH16];
#pragma unroll
    for (int s = 0; s < CH16; ++s) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        x[e] = *reinterpret_cast<const float *>(fb + (size_t)(16 * s + e) * P * 4u + a_off);  // uniform base + lane offset
#pragma unroll
      for (int e = 0; e < 8; ++e) a[s][e] = (__bf16)x[e];
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < CH16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], bw[s], acc, 0, 0, 0);
    if (j < J) {
      char *ob = reinterpret_cast<char *>(out + (size_t)n0 * J);
#pragma unroll
      for (int e = 0; e < 16; ++e)
        *reinterpret_cast<float *>(ob + (size_t)((e & 3) + 8 * (e >> 2)) * J * 4u + o_off) = acc[e];
    }
  }
}

static float bf16_round(float x) {  // round to nearest even, as v_cvt_pk_bf16_f32
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  memcpy(&x, &u, 4);
  return x;
}

int main() {
  const int C = 64, J = 128, P = 16 * 4096;
  std::vector<float> F((size_t)C * P), W((size_t)J * C);
  srand(1);
  for (auto &v : F) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto &v : W) v = (float)rand() / RAND_MAX - 0.5f;
  float *dF, *dW, *dO;
  hipMalloc(&dF, F.size() * 4);
  hipMalloc(&dW, W.size() * 4);
  hipMalloc(&dO, (size_t)P * J * 4);
  hipMemcpy(dF, F.data(), F.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  const int nblocks = P / 32;
  const dim3 grid(nblocks < 256 ? nblocks : 256, (J / 32 + 3) / 4);
  auto launch = [&] { hipLaunchKernelGGL(skinny_fwd_bf16<4>, grid, dim3(256), 0, 0, dF, dW, dO, C, P, J, nblocks); };
  launch();
  hipDeviceSynchronize();
  std::vector<float> O((size_t)P * J);
  hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0.0, scale = 0.0;
  for (int p = 0; p < P; p += 997)
    for (int j = 0; j < J; ++j) {
      double want = 0.0;
      for (int c = 0; c < C; ++c) want += (double)bf16_round(F[(size_t)c * P + p]) * (double)bf16_round(W[(size_t)j * C + c]);
      worst = fmax(worst, fabs(want - (double)O[(size_t)p * J + j]));
      scale = fmax(scale, fabs(want));
    }
  printf("worst error %.3e of %.3e (relative %.2e; f32 accumulation of 64 exact products: ~1e-6 expected)\n", worst, scale, worst / scale);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(e0, 0);
  const int reps = 50;
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1000.0 / reps;
  printf("bf16 LDS-free forward product [%d x %d] x [%d x %d]: %.1f us per launch, %.2f TB/s of (F in + rows out)\n", P, C, C, J, us,
         ((double)C * P * 4 + (double)P * J * 4) / us / 1e6);
  return 0;
}

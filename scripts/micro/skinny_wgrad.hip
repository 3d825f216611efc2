// Prototype: the PointWiseMLP's weight-gradient product  dwcat[o][c] = sum_p dght[p][o] * F[c][p]  (dght [P][128]
// point-major, F [64][P] channel-major, P = 65536) as K-slices with NO operand staging through LDS: both operands' MFMA
// fragments come straight from global memory -- a dght row's 32 consecutive outputs are the lanes of an A fragment, four
// consecutive points of a channel (one 16-byte load) feed four MFMA steps of a B fragment -- each wave keeps two 32 x 32
// output tiles, a workgroup of four waves the whole 128 x 64 output of its slice of points.
//   hipcc --offload-arch=gfx950 -O3 -o skinny_wgrad skinny_wgrad.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int C = 64, J = 128;

__global__ __launch_bounds__(256) void skinny_wgrad(const float *__restrict__ dght, const float *__restrict__ F,
                                                    float *__restrict__ partial, int P, int L) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int it = wave & 1, jt = wave >> 1;
  const int p_lo = blockIdx.x * L, p_hi = p_lo + L;
  f32x16 acc[2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[x][e] = 0.f;
  const float *frow = F + (size_t)(32 * jt + lr) * P + 4 * lh;
  const float *arow = dght + (size_t)(4 * lh) * J + 32 * (2 * it) + lr;
  float4 b_cur = *reinterpret_cast<const float4 *>(frow + p_lo), b_nxt = b_cur;
  float a_cur[2][4], a_nxt[2][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int x = 0; x < 2; ++x) a_cur[x][t] = arow[(size_t)(p_lo + t) * J + 32 * x];
  for (int p0 = p_lo; p0 < p_hi; p0 += 8) {
    const int pn = p0 + 8 < p_hi ? p0 + 8 : p0;
    b_nxt = *reinterpret_cast<const float4 *>(frow + pn);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int x = 0; x < 2; ++x) a_nxt[x][t] = arow[(size_t)(pn + t) * J + 32 * x];
    const float bb[4] = {b_cur.x, b_cur.y, b_cur.z, b_cur.w};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int x = 0; x < 2; ++x) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[x][t], bb[t], acc[x], 0, 0, 0);
    b_cur = b_nxt;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int x = 0; x < 2; ++x) a_cur[x][t] = a_nxt[x][t];
  }
  float *out = partial + (size_t)blockIdx.x * J * C;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = 32 * (2 * it + x) + (e & 3) + 8 * (e >> 2) + 4 * lh;
      out[(size_t)i * C + 32 * jt + lr] = acc[x][e];
    }
}

int main() {
  const int P = 65536;
  std::vector<float> hD((size_t)P * J), hF((size_t)C * P);
  srand(5);
  for (auto &x : hD) x = (rand() % 2001 - 1000) / 1000.f;
  for (auto &x : hF) x = (rand() % 2001 - 1000) / 1000.f;
  float *D, *F, *part;
  hipMalloc(&D, hD.size() * 4);
  hipMalloc(&F, hF.size() * 4);
  hipMalloc(&part, (size_t)1024 * J * C * 4);
  hipMemcpy(D, hD.data(), hD.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(F, hF.data(), hF.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int slices = 128; slices <= 1024; slices *= 2) {
    const int L = P / slices;
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL(skinny_wgrad, dim3(slices), dim3(256), 0, 0, D, F, part, P, L);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      if (rep > 0 && ms < best) best = ms;
    }
    printf("%4d slices of %4d points: %6.1f us  (%.1f TFLOP/s)\n", slices, L, best * 1e3, 2.0 * P * C * J / best / 1e9);
  }
  const int slices = 1024, L = P / slices;
  std::vector<float> hp((size_t)slices * J * C);
  hipMemcpy(hp.data(), part, hp.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int t = 0; t < 500; ++t) {
    const int s = rand() % slices, o = rand() % J, c = rand() % C;
    double r = 0;
    for (int p = s * L; p < (s + 1) * L; ++p) r += (double)hD[(size_t)p * J + o] * hF[(size_t)c * P + p];
    worst = fmax(worst, fabs(r - hp[((size_t)s * J + o) * C + c]));
  }
  printf("max |error| on 500 sampled partial outputs: %.3e\n", worst);
  return 0;
}

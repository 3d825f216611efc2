// abi_host.cpp -- libcl3d.so used from plain C++ through include/cl3d.h: no Python, no torch.
// Device memory comes from hipMalloc, the work is enqueued on a HIP stream the caller owns, errors are return
// codes + cl3d_last_error_string().  Runs the reference-visible boundary (masked ordered ball query -> gather ->
// scatter-add) on a random cloud and checks two invariants that need no oracle:
//   * every returned neighbour lies inside the radius and rows are sorted by distance;
//   * <gather(f), a> == <f, scatter(a)>  (the scatter-add is the adjoint of the gather).
// Build + run (tests/test_abi_host_gpu.py does exactly this on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/abi_host.cpp -Lcloserlook3d_amd -lcl3d \
//         -Wl,-rpath,$PWD/closerlook3d_amd -o /tmp/abi_host && /tmp/abi_host
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cl3d.h"

#define HIP_OK(x)                                                              \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
      return 2;                                                                \
    }                                                                          \
  } while (0)
#define CL3D_OK_OR_DIE(x)                                                      \
  do {                                                                         \
    if ((x) != 0) {                                                            \
      std::fprintf(stderr, "%s: %s\n", #x, cl3d_last_error_string());         \
      return 3;                                                                \
    }                                                                          \
  } while (0)

int main() {
  const int B = 4, N = 4096, M = N, K = 32, C = 16;
  const float radius = 0.14f;
  std::vector<float> xyz((size_t)B * N * 3), feat((size_t)B * C * N), a((size_t)B * C * M * K);
  std::vector<int> mask((size_t)B * N, 1);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
  for (auto &v : xyz) v = rnd();
  for (auto &v : feat) v = rnd() - 0.5f;
  for (auto &v : a) v = rnd() - 0.5f;

  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  float *d_xyz, *d_feat, *d_a, *d_g, *d_sc;
  int *d_mask, *d_idx, *d_idxm;
  HIP_OK(hipMalloc(&d_xyz, xyz.size() * 4));
  HIP_OK(hipMalloc(&d_feat, feat.size() * 4));
  HIP_OK(hipMalloc(&d_a, a.size() * 4));
  HIP_OK(hipMalloc(&d_g, a.size() * 4));
  HIP_OK(hipMalloc(&d_sc, feat.size() * 4));
  HIP_OK(hipMalloc(&d_mask, mask.size() * 4));
  HIP_OK(hipMalloc(&d_idx, (size_t)B * M * K * 4));
  HIP_OK(hipMalloc(&d_idxm, (size_t)B * M * K * 4));
  HIP_OK(hipMemcpyAsync(d_xyz, xyz.data(), xyz.size() * 4, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemcpyAsync(d_feat, feat.data(), feat.size() * 4, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemcpyAsync(d_a, a.data(), a.size() * 4, hipMemcpyHostToDevice, st));
  HIP_OK(hipMemcpyAsync(d_mask, mask.data(), mask.size() * 4, hipMemcpyHostToDevice, st));

  const size_t ws_bytes = cl3d_workspace_bytes(CL3D_OP_BALL_QUERY, B, N, M, K, 0);
  void *ws = nullptr;
  if (ws_bytes) HIP_OK(hipMalloc(&ws, ws_bytes));
  CL3D_OK_OR_DIE(cl3d_masked_ordered_ball_query(d_xyz, d_xyz, d_mask, d_mask, B, M, N, radius, K, d_idx, d_idxm, ws,
                                                ws_bytes, st));
  CL3D_OK_OR_DIE(cl3d_group_points(d_feat, d_idx, B, C, N, M, K, d_g, st));
  CL3D_OK_OR_DIE(cl3d_group_points_grad(d_a, d_idx, B, C, N, M, K, d_sc, nullptr, 0, st));

  std::vector<int> idx((size_t)B * M * K), idxm(idx.size());
  std::vector<float> g(a.size()), sc(feat.size());
  HIP_OK(hipMemcpyAsync(idx.data(), d_idx, idx.size() * 4, hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(idxm.data(), d_idxm, idxm.size() * 4, hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(g.data(), d_g, g.size() * 4, hipMemcpyDeviceToHost, st));
  HIP_OK(hipMemcpyAsync(sc.data(), d_sc, sc.size() * 4, hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));

  // (1) radius and order
  long long bad = 0;
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < M; ++j) {
      const float *q = &xyz[((size_t)b * N + j) * 3];
      float prev = -1.f;
      for (int k = 0; k < K; ++k) {
        const size_t e = ((size_t)b * M + j) * K + k;
        if (!idxm[e]) break;  // wrap-around padding follows the real neighbours
        const float *p = &xyz[((size_t)b * N + idx[e]) * 3];
        const float d2 = (q[0] - p[0]) * (q[0] - p[0]) + (q[1] - p[1]) * (q[1] - p[1]) + (q[2] - p[2]) * (q[2] - p[2]);
        if (!(d2 < radius * radius * 1.0001f) || d2 + 1e-7f < prev) ++bad;
        prev = d2;
      }
    }
  // (2) adjointness in double
  double lhs = 0.0, rhs = 0.0;
  for (size_t i = 0; i < g.size(); ++i) lhs += (double)g[i] * a[i];
  for (size_t i = 0; i < sc.size(); ++i) rhs += (double)sc[i] * feat[i];
  const double rel = std::fabs(lhs - rhs) / (std::fabs(lhs) + 1e-30);
  std::printf("abi v%d: %d clouds x %d points, K=%d: order/radius violations %lld, adjoint mismatch %.2e\n",
              cl3d_abi_version(), B, N, K, bad, rel);
  // (3) error path: a bad argument is a return code with a message, never an abort
  const int rc = cl3d_group_points(nullptr, d_idx, B, C, N, M, K, d_g, st);
  std::printf("null input -> rc %d (%s)\n", rc, cl3d_last_error_string());
  return (bad == 0 && rel < 1e-6 && rc != 0) ? 0 : 1;
}

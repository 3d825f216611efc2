// reduce_pass_host.cpp -- a PosPool (xyz, sum) LocalAggregation pass from plain C++ through the round-6 pass calls of
// include/cl3d.h (cl3d_reduce_train_forward / _backward, csrc/pass.hip): no Python, no torch; every buffer the caller's.
// Small enough to check against a literal restatement ON THE HOST: the library's own idx (copied back) drives a plain
// triple loop  out[b,c,j] = sum_k rel[c % 3](j,k) * f[b,c,idx[j,k]] * mask  (local_aggregation_operators.py:65-69,99-103)
// and its adjoint; four steps with one argument block walk the launch-graph path (direct, captured, replayed, replayed)
// and must give the first step's bits.
// Build + run (tests/test_abi_host_gpu.py does this on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/reduce_pass_host.cpp -Lcloserlook3d_amd -lcl3d \
//         -Wl,-rpath,$PWD/closerlook3d_amd -o /tmp/reduce_pass_host && /tmp/reduce_pass_host
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "cl3d.h"

#define HIP_OK(x)                                                   \
  do {                                                              \
    hipError_t e_ = (x);                                            \
    if (e_ != hipSuccess) {                                         \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      return 2;                                                     \
    }                                                               \
  } while (0)
#define CL3D_OK_OR_DIE(x)                                                 \
  do {                                                                    \
    if ((x) != 0) {                                                       \
      std::fprintf(stderr, "%s: %s\n", #x, cl3d_last_error_string());    \
      return 3;                                                           \
    }                                                                     \
  } while (0)

template <class T>
static T *dev(size_t n) {
  void *p = nullptr;
  return hipMalloc(&p, (n ? n : 1) * sizeof(T)) == hipSuccess ? static_cast<T *>(p) : nullptr;
}

int main() {
  const int B = 2, N = 1024, M = N, K = 16, C = 12;
  const float radius = 0.2f;
  unsigned s = 77u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
  std::vector<float> xyz((size_t)B * N * 3), feat((size_t)B * C * N), gout((size_t)B * C * M);
  std::vector<int> mask((size_t)B * N, 1);
  for (auto &v : xyz) v = rnd();
  for (auto &v : feat) v = rnd() - 0.5f;
  for (auto &v : gout) v = rnd() - 0.5f;

  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  long long captures0 = 0, replays0 = 0;
  CL3D_OK_OR_DIE(cl3d_pwmlp_pass_graph_stats(&captures0, &replays0));
  cl3d_reduce_pass p;
  std::memset(&p, 0, sizeof(p));
  p.B = B; p.N = N; p.M = M; p.K = K; p.C = C;
  p.op = 0;            // PosPool, position_embedding 'xyz'
  p.normalize = 1;     // rel / radius (MaskedQueryAndGroup normalize_xyz=True, local_aggregation_operators.py:36)
  p.reduction = 0;     // 'sum'
  p.radius = radius;
  p.nparts = cl3d_fused_param_partials(p.op, B, N, C);  // 0: PosPool has no parameters in front of its output transform
  p.bq_ws_bytes = cl3d_workspace_bytes(CL3D_OP_BALL_QUERY, B, N, M, K, 0);
  p.csr_ws_bytes = cl3d_workspace_bytes(CL3D_OP_INVERSE_INDEX, B, N, M * K, 1, 0);
  float *d_xyz = dev<float>(xyz.size()), *d_feat = dev<float>(feat.size()), *d_gout = dev<float>(gout.size());
  int *d_mask = dev<int>(mask.size());
  HIP_OK(hipMemcpy(d_xyz, xyz.data(), xyz.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_feat, feat.data(), feat.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_gout, gout.data(), gout.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_mask, mask.data(), mask.size() * 4, hipMemcpyHostToDevice));
  p.query_xyz = p.support_xyz = d_xyz;
  p.query_mask = p.support_mask = d_mask;
  p.features = d_feat;
  p.idx = dev<int32_t>((size_t)B * M * K); p.idx_mask = dev<int32_t>((size_t)B * M * K);
  p.inv_off = dev<int32_t>((size_t)B * (N + 1)); p.inv_slots = dev<int32_t>((size_t)B * M * K);
  p.bq_ws = dev<char>(p.bq_ws_bytes); p.csr_ws = dev<char>(p.csr_ws_bytes);
  p.ft = dev<float>((size_t)B * N * C); p.out = dev<float>((size_t)B * C * M); p.slotrec = dev<float>((size_t)B * M * K * 4);
  p.gout = d_gout; p.gout_t = dev<float>((size_t)B * M * C); p.dfeat = dev<float>((size_t)B * C * N);

  std::vector<float> out((size_t)B * C * M), dfeat((size_t)B * C * N), out1, dfeat1;
  int differing = 0;
  for (int step = 0; step < 4; ++step) {
    CL3D_OK_OR_DIE(cl3d_reduce_train_forward(&p, st));
    CL3D_OK_OR_DIE(cl3d_reduce_train_backward(&p, st));
    HIP_OK(hipMemcpyAsync(out.data(), p.out, out.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(dfeat.data(), p.dfeat, dfeat.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    if (step == 0) {
      out1 = out; dfeat1 = dfeat;
    } else {
      differing += std::memcmp(out.data(), out1.data(), out.size() * 4) != 0;
      differing += std::memcmp(dfeat.data(), dfeat1.data(), dfeat.size() * 4) != 0;
    }
  }
  // host restatement from the library's own neighbour lists
  std::vector<int32_t> idx((size_t)B * M * K), idx_mask((size_t)B * M * K);
  HIP_OK(hipMemcpy(idx.data(), p.idx, idx.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(idx_mask.data(), p.idx_mask, idx_mask.size() * 4, hipMemcpyDeviceToHost));
  std::vector<double> want_out((size_t)B * C * M, 0.0), want_df((size_t)B * C * N, 0.0);
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < M; ++j)
      for (int k = 0; k < K; ++k) {
        const size_t e = ((size_t)b * M + j) * K + k;
        const int i = idx[e];
        const double w = idx_mask[e] ? 1.0 : 0.0;  // (every query is valid here: the mask is idx_mask alone, :99-101)
        for (int c = 0; c < C; ++c) {
          const int a = c % 3;
          const double rel = ((double)xyz[((size_t)b * N + i) * 3 + a] - (double)xyz[((size_t)b * M + j) * 3 + a]) / radius;
          want_out[((size_t)b * C + c) * M + j] += w * rel * feat[((size_t)b * C + c) * N + i];
          want_df[((size_t)b * C + c) * N + i] += w * rel * gout[((size_t)b * C + c) * M + j];
        }
      }
  double worst_out = 0.0, worst_df = 0.0, scale_out = 1e-30, scale_df = 1e-30;
  for (size_t e = 0; e < want_out.size(); ++e) {
    worst_out = std::fmax(worst_out, std::fabs(want_out[e] - out1[e]));
    scale_out = std::fmax(scale_out, std::fabs(want_out[e]));
  }
  for (size_t e = 0; e < want_df.size(); ++e) {
    worst_df = std::fmax(worst_df, std::fabs(want_df[e] - dfeat1[e]));
    scale_df = std::fmax(scale_df, std::fabs(want_df[e]));
  }
  long long captures = 0, replays = 0;
  CL3D_OK_OR_DIE(cl3d_pwmlp_pass_graph_stats(&captures, &replays));
  captures -= captures0; replays -= replays0;
  std::printf("PosPool pass calls from C++: %d clouds x %d points, K=%d, C=%d: output error %.2e, feature-gradient error %.2e "
              "(relative to the largest value), steps differing from step 1: %d, passes captured %lld, replayed %lld\n",
              B, N, K, C, worst_out / scale_out, worst_df / scale_df, differing, captures, replays);
  const bool ok = worst_out / scale_out < 1e-5 && worst_df / scale_df < 1e-5 && differing == 0 && captures == 2 && replays >= 4;
  return ok ? 0 : 1;
}

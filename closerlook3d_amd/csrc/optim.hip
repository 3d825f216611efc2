// optim.hip -- the SGD update over flat parameter / gradient / momentum buffers in one launch.
//
// The reference's training loops update with torch.optim.SGD (function/train_modelnet_dist.py:137-141,
// function/train_s3dis_dist.py and train_partnet_dist.py likewise: momentum, weight decay from the YAML).  On the
// device that optimizer is already one multi-tensor kernel; what this entry point removes is the HOST side of an eagerly
// launched step -- torch.optim.SGD.step + zero_grad are ~0.11 ms of Python per step next to a 0.29 ms step
// (profiles/r05/eager_host.txt) -- for callers that keep their parameters in one flat buffer (closerlook3d_amd/optim.py:
// FlatSGD re-points every parameter at a view of it; measured on the repository's benches: no gain, see there).
// torch.optim.SGD's arithmetic, per element:
//     g = grad + weight_decay * p;   buf = first step ? g : momentum * buf + (1 - dampening) * g;
//     g = nesterov ? g + momentum * buf : buf   (momentum != 0);      p -= lr * g
// and, optionally, grad = 0 for the next step's accumulation (the flat gradient buffer is what autograd adds into).
#include "cl3d_common.h"

namespace cl3d {

struct SgdArgs {
  float *p;
  float *g;
  float *buf;  // null: no momentum
  long long n;
  float lr, momentum, dampening, weight_decay;
  int nesterov, first_step, zero_grad;
};

__device__ __forceinline__ float sgd_one(const SgdArgs &a, float p, float g, float &buf) {
  if (a.weight_decay != 0.f) g = __builtin_fmaf(a.weight_decay, p, g);
  if (a.buf != nullptr) {
    buf = a.first_step ? g : __builtin_fmaf(a.momentum, buf, (1.f - a.dampening) * g);
    g = a.nesterov ? __builtin_fmaf(a.momentum, buf, g) : buf;
  }
  return __builtin_fmaf(-a.lr, g, p);
}

template <bool VEC4>
__global__ __launch_bounds__(256) void sgd_step_kernel(SgdArgs a) {
  const long long stride = (long long)gridDim.x * 256;
  if (VEC4) {
    const long long n4 = a.n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      float4 p = reinterpret_cast<float4 *>(a.p)[i];
      const float4 g = reinterpret_cast<const float4 *>(a.g)[i];
      float4 b = a.buf != nullptr && !a.first_step ? reinterpret_cast<float4 *>(a.buf)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      p.x = sgd_one(a, p.x, g.x, b.x); p.y = sgd_one(a, p.y, g.y, b.y);
      p.z = sgd_one(a, p.z, g.z, b.z); p.w = sgd_one(a, p.w, g.w, b.w);
      reinterpret_cast<float4 *>(a.p)[i] = p;
      if (a.buf != nullptr) reinterpret_cast<float4 *>(a.buf)[i] = b;
      if (a.zero_grad) reinterpret_cast<float4 *>(a.g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {  // the last n % 4
      float b = a.buf != nullptr && !a.first_step ? a.buf[i] : 0.f;
      a.p[i] = sgd_one(a, a.p[i], a.g[i], b);
      if (a.buf != nullptr) a.buf[i] = b;
      if (a.zero_grad) a.g[i] = 0.f;
    }
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
      float b = a.buf != nullptr && !a.first_step ? a.buf[i] : 0.f;
      a.p[i] = sgd_one(a, a.p[i], a.g[i], b);
      if (a.buf != nullptr) a.buf[i] = b;
      if (a.zero_grad) a.g[i] = 0.f;
    }
  }
}

}  // namespace cl3d

extern "C" int cl3d_sgd_step(float *param, float *grad, float *momentum_buf, long long n, float lr, float momentum,
                             float dampening, float weight_decay, int nesterov, int first_step, int zero_grad,
                             cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(n >= 0, "sgd_step: bad size");
  if (n == 0) return CL3D_OK;
  CL3D_REQUIRE(param && grad, "sgd_step: null pointer");
  CL3D_REQUIRE((momentum != 0.f) == (momentum_buf != nullptr), "sgd_step: a momentum buffer goes with momentum != 0");
  CL3D_REQUIRE(!nesterov || (momentum > 0.f && dampening == 0.f), "sgd_step: Nesterov needs momentum > 0 and no dampening");
  SgdArgs a{param, grad, momentum_buf, n, lr, momentum, dampening, weight_decay, nesterov, first_step, zero_grad};
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  const bool vec = al16(param) && al16(grad) && (momentum_buf == nullptr || al16(momentum_buf));
  long long items = vec ? (n + 3) / 4 : n;
  long long grid = (items + 255) / 256;
  if (grid > 4096) grid = 4096;
  if (vec) hipLaunchKernelGGL(sgd_step_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(sgd_step_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("cl3d_sgd_step");
}

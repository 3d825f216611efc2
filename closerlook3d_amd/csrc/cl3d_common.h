// cl3d_common.h -- shared device/host helpers of libcl3d (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/cl3d.h"

#define CL3D_WAVE 64

namespace cl3d {

// ---- per-thread error text ------------------------------------------------------------
inline char *err_buf() {
  static thread_local char buf[512] = "";
  return buf;
}
// (format-checked: a bare '%' in a message is a conversion to vsnprintf -- one such message crashed its own
// bad-argument test in round 3)
inline int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CL3D_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return CL3D_OK;
}
#define CL3D_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) return cl3d::fail(CL3D_E_INVALID, __VA_ARGS__); \
  } while (0)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Dynamic LDS above 64 KiB has to be granted per kernel AND per device (the code object is loaded once per
// device).  `granted` is the call site's own bit mask of devices already done; safe to race (idempotent).
inline int lds_opt_in(std::atomic<unsigned long long> &granted, const void *kernel, size_t bytes, const char *who) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(CL3D_E_LAUNCH, "%s: no current device", who);
  const bool tracked = dev >= 0 && dev < 64;
  if (tracked && ((granted.load(std::memory_order_relaxed) >> dev) & 1ull)) return CL3D_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return fail(CL3D_E_LAUNCH, "%s: LDS opt-in (%zu B): %s", who, bytes, hipGetErrorString(e));
  if (tracked) granted.fetch_or(1ull << dev, std::memory_order_relaxed);
  return CL3D_OK;
}

// ---- ticket counters of the in-launch sums (the K slices of a product: csrc/mfma_gemm.hip; the partial blocks of a
// BatchNorm statistics pass: csrc/bn_relu.hip): one zero-initialised ring per device, handed out in pieces of one
// counter per output tile / channel.  A piece is zero again when its launch has finished (the last arrival resets it).  Launches
// that are being CAPTURED into a HIP graph keep their piece for as long as the graph lives (every replay uses it), so
// they draw from the upper half of the ring, which is never handed out twice -- when it is used up a captured launch gets
// no piece and takes the two-launch form; eager launches draw from the lower half, round and round (2^19 counters: far
// more than the tiles of all launches that can be in flight at once -- a sliced product has few tiles, that is why it was
// sliced).  The ring is allocated outside stream capture (hipMalloc + hipMemset are not stream operations); a first
// call that arrives during a capture gets no piece.
constexpr size_t kTicketRing = (size_t)1 << 20;
inline unsigned *ticket_piece(size_t n, hipStream_t st) {
  static std::mutex mu;
  static unsigned *ring[64] = {};
  static size_t next_eager[64] = {}, next_captured[64] = {};
  int dev = 0;
  if (n == 0 || n > kTicketRing / 8 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  const bool capturing = cs != hipStreamCaptureStatusNone;
  std::lock_guard<std::mutex> lock(mu);
  if (ring[dev] == nullptr) {
    if (capturing) return nullptr;
    unsigned *p = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&p), kTicketRing * sizeof(unsigned)) != hipSuccess ||
        hipMemset(p, 0, kTicketRing * sizeof(unsigned)) != hipSuccess) {
      (void)hipGetLastError();
      if (p) (void)hipFree(p);
      return nullptr;
    }
    ring[dev] = p;
  }
  const size_t half = kTicketRing / 2;
  if (capturing) {
    if (next_captured[dev] + n > half) return nullptr;
    unsigned *piece = ring[dev] + half + next_captured[dev];
    next_captured[dev] += n;
    return piece;
  }
  if (next_eager[dev] + n > half) next_eager[dev] = 0;
  unsigned *piece = ring[dev] + next_eager[dev];
  next_eager[dev] += n;
  return piece;
}

// last arrival at a ticket: called by EVERY thread of a workgroup after its device-scope stores (each wave waits for its
// stores to be acknowledged -- the explicit s_waitcnt: a workgroup-scope barrier alone need not wait for vmcnt -- and the
// ticket is a device-scope atomic issued behind the barrier); block-uniform result.  The last arrival
// puts the counter back to zero (`total` workgroups draw from `ticket`).  s_flag: one int of LDS.
__device__ __forceinline__ bool last_arrival(unsigned *ticket, unsigned total, int *s_flag) {
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned drawn = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = drawn == total - 1u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
}

// ---- device helpers -------------------------------------------------------------------
// Squared distance in the canonical operation order (see DESIGN.md "floating-point canon").  Form 0 is what
// hipcc -O2 makes of the reference expression (masked_ordered_ball_query_gpu.cu:56-57) on gfx950,
//   d2 = fadd(fma(dy,dy, fmul(dx,dx)), fmul(dz,dz)),
// and is the default.  CL3D_D2_FORM (a build-time switch, same numbering as the oracle's) selects the two other
// plausible contractions, for a maintainer who has to match indices produced by another compiler:
//   1 = no contraction, 2 = the full left-to-right fma chain (what nvcc normally emits).
// The library is compiled with -ffp-contract=off, so nothing here is re-fused.
#ifndef CL3D_D2_FORM
#define CL3D_D2_FORM 0
#endif
__device__ __forceinline__ float dist2(float qx, float qy, float qz, float x, float y, float z) {
  const float dx = qx - x, dy = qy - y, dz = qz - z;
#if CL3D_D2_FORM == 0
  const float xx = dx * dx;
  const float zz = dz * dz;
  return __builtin_fmaf(dy, dy, xx) + zz;
#elif CL3D_D2_FORM == 1
  const float xx = dx * dx;
  const float yy = dy * dy;
  const float zz = dz * dz;
  return (xx + yy) + zz;
#else
  const float xx = dx * dx;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, xx));
#endif
}

// t / K for slot ids (t < 2^24, K <= 255) without the ~20-instruction integer division: magic = ceil(2^32 / K);
// magic == 0 selects the plain division (anything larger)
__device__ __forceinline__ int div_k(int t, unsigned magic, int K) {
  return K == 1 ? t : (magic != 0u ? (int)__umulhi((unsigned)t, magic) : t / K);
}
inline unsigned div_magic(int K, long long max_t = 0) {
  if (K <= 1 || K > 255 || max_t >= (1ll << 24)) return 0u;
  return (unsigned)((0x100000000ull + (unsigned)K - 1) / (unsigned)K);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// number of set bits of m strictly below this lane
__device__ __forceinline__ int prefix_popc(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    unsigned long long o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}

// first index i in [0,n) with mask[i]==0, else n.  All threads of the block call it;
// result is block-uniform.  `s_tmp` is one int of LDS.
__device__ __forceinline__ int block_first_zero(const int *__restrict__ mask, int n, int *s_tmp) {
  if (threadIdx.x == 0) *s_tmp = n;
  __syncthreads();
  int best = n;
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    if (mask[i] == 0) { best = i; break; }
  if (best < n) atomicMin(s_tmp, best);
  __syncthreads();
  int r = *s_tmp;
  __syncthreads();
  return r;
}

}  // namespace cl3d

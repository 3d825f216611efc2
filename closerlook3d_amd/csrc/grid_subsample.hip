// grid_subsample.hip -- masked grid subsampling for gfx950.
//
// Replaces masked_grid_subsampling_gpu.cu:11-153 of the reference, which runs ONE thread per
// cloud (<<<b,1>>>): three linear passes and two in-thread sorts of N keys, serial.
// Same results, bit for bit, from one 1024-thread workgroup per cloud:
//
//   1. bounding box over all N points (reference :31-46 does not look at the mask here) and the
//      number of leading valid points -- block reductions;
//   2. one 64-bit key per valid point, (cell id biased to unsigned) << 32 | original index, sorted
//      by a bitonic network in LDS.  Keys are unique, so the result equals the reference's stable
//      sort by cell id;
//   3. every thread walks a contiguous run of the sorted array; cell heads are counted, an
//      exclusive block scan gives each head its rank (= the reference's `top`), and the thread
//      that owns a head folds that cell's points in ascending original index with plain float
//      adds -- the same summation order as the reference's sequential loop (:84-122);
//   4. the reference's LCG shuffle (keys k0 = cell0 % 256, k_i = (17 k_{i-1} + 139) % 256, stable
//      sort, :125-135) has period 256 after a transient of < 256 steps, so the position of
//      barycentre i in the shuffled order is computed in closed form from a 512-entry table
//      instead of sorting; each barycentre is written straight to its final slot;
//   5. wrap-around padding with mask 0 (:146-151).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "ball_query.h"

namespace cl3d {

constexpr int kSubThreads = 1024;
constexpr int kSubMaxN = 16384;  // 128 KiB of sort keys; the CU has 160 KiB of LDS

__device__ __forceinline__ int key_cell(unsigned long long k) {
  return (int)(((unsigned)(k >> 32)) ^ 0x80000000u);
}
__device__ __forceinline__ int key_orig(unsigned long long k) { return (int)(unsigned)(k & 0xffffffffull); }

// Large clouds (N > kSubMaxN): the (cell, index) keys do not fit LDS.  A first kernel writes them to HBM as
// {cloud:6 | biased cell:32 | index:26}, one device-wide radix sort (rocPRIM) orders all clouds at once, and
// the same per-cloud kernel runs with PRESORTED = true, reading its cloud's slice instead of sorting in LDS.
// (The index field is deliberately wider than 24 bits: hipcc 7.2 for gfx950 drops an `x & 0xffffff` in front
// of an address multiply -- `p[(k & 0xffffff) * 3]` compiles to v_mad_u64_u32 on the unmasked dword -- which
// showed up here as a memory fault; masks wider than 24 bits are compiled correctly.)
constexpr int kBigIdxBits = 26;
__device__ __forceinline__ unsigned long long big_key(int b, int cell, int i, bool valid) {
  const unsigned long long c = valid ? (unsigned long long)(((unsigned)cell) ^ 0x80000000u) : 0xffffffffull;
  return ((unsigned long long)(unsigned)b << (32 + kBigIdxBits)) | (c << kBigIdxBits) | (unsigned long long)(unsigned)i;
}

struct SubParams {
  int nv;
  int pad[3];
};

template <bool PRESORTED, bool KEYS_ONLY>
__global__ __launch_bounds__(kSubThreads) void grid_subsample_kernel(
    const float *__restrict__ xyz, const int *__restrict__ mask, int N, int m, float dl, int P,
    float *__restrict__ sub_xyz, int *__restrict__ sub_mask, unsigned long long *__restrict__ gkeys,
    SubParams *__restrict__ params) {
  extern __shared__ unsigned long long lds_keys[];  // [P] (in-LDS path only)
  __shared__ float s_red[6][kSubThreads / 64];
  __shared__ int s_scan[kSubThreads / 64];
  __shared__ int s_scan2[8];
  __shared__ int s_tmp;
  __shared__ int s_T[512];
  __shared__ int s_E[256];
  __shared__ int s_invB[256];
  __shared__ int s_cntA[511];
  __shared__ int s_base[512];

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const float *p = xyz + (size_t)b * N * 3;
  const int *mk = mask + (size_t)b * N;
  float *o = sub_xyz + (size_t)b * m * 3;
  int *om = sub_mask + (size_t)b * m;

  const unsigned long long *skeys = PRESORTED ? gkeys + (size_t)b * N : lds_keys;
  auto key_at = [&](int pos) -> unsigned long long {
    if constexpr (PRESORTED) {  // strip the cloud byte, bring {cell, index} back to the 32|32 layout
      const unsigned long long k = skeys[pos];
      return (((k >> kBigIdxBits) & 0xffffffffull) << 32) | (k & ((1ull << kBigIdxBits) - 1));
    } else {
      return skeys[pos];
    }
  };
  int nv;
  float mn[3], mx[3];
  if constexpr (PRESORTED) {
    nv = params[b].nv;
  } else {
  nv = block_first_zero(mk, N, &s_tmp);

  // ---- 1. bounding box over all N points
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mn[a] = p[a];
    mx[a] = p[a];
  }
  for (int i = tid; i < N; i += kSubThreads) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = p[i * 3 + a];
      if (v > mx[a]) mx[a] = v;
      if (v < mn[a]) mn[a] = v;
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float omx = __shfl_xor(mx[a], off, 64);
      const float omn = __shfl_xor(mn[a], off, 64);
      if (omx > mx[a]) mx[a] = omx;
      if (omn < mn[a]) mn[a] = omn;
    }
    if (lane == 0) {
      s_red[a][wave] = mn[a];
      s_red[3 + a][wave] = mx[a];
    }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mn[a] = s_red[a][0];
    mx[a] = s_red[3 + a][0];
    for (int w = 1; w < kSubThreads / 64; ++w) {
      const float vmn = s_red[a][w], vmx = s_red[3 + a][w];
      if (vmn < mn[a]) mn[a] = vmn;
      if (vmx > mx[a]) mx[a] = vmx;
    }
  }
  const float inv = 1.0f / dl;
  const float ox = floorf(mn[0] * inv) * dl;
  const float oy = floorf(mn[1] * inv) * dl;
  const float oz = floorf(mn[2] * inv) * dl;
  const int NX = (int)floorf((mx[0] - ox) / dl) + 1;
  const int NY = (int)floorf((mx[1] - oy) / dl) + 1;

  // ---- 2. keys + bitonic sort
  for (int i = tid; i < P; i += kSubThreads) {
    unsigned long long key = ~0ull;
    if (i < nv) {
      const int iX = (int)floorf((p[i * 3 + 0] - ox) / dl);
      const int iY = (int)floorf((p[i * 3 + 1] - oy) / dl);
      const int iZ = (int)floorf((p[i * 3 + 2] - oz) / dl);
      const int cell = iX + NX * iY + NX * NY * iZ;
      key = ((unsigned long long)(((unsigned)cell) ^ 0x80000000u) << 32) | (unsigned)i;
    }
    if constexpr (KEYS_ONLY) {
      if (i < N) gkeys[(size_t)b * N + i] = big_key(b, (int)(((unsigned)(key >> 32)) ^ 0x80000000u), i, i < nv);
    } else {
      lds_keys[i] = key;
    }
  }
  if constexpr (KEYS_ONLY) {
    if (tid == 0) params[b].nv = nv;
    return;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (P >> 1); t += kSubThreads) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const unsigned long long a = lds_keys[i], c = lds_keys[l];
        const bool up = (i & k) == 0;
        if ((a > c) == up) {
          lds_keys[i] = c;
          lds_keys[l] = a;
        }
      }
      __syncthreads();
    }
  }

  }  // !PRESORTED

  // ---- 3. heads: contiguous run per thread, block scan of head counts
  const int run = (P + kSubThreads - 1) / kSubThreads;
  const int r0 = tid * run;
  int r1 = r0 + run;
  r1 = r1 < nv ? r1 : nv;
  int heads = 0;
  for (int pos = r0; pos < r1; ++pos)
    heads += (pos == 0 || key_cell(key_at(pos)) != key_cell(key_at(pos - 1))) ? 1 : 0;
  // exclusive scan over threads
  int incl = heads;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) s_scan[wave] = incl;
  __syncthreads();
  int wave_off = 0, total = 0;
  for (int w = 0; w < kSubThreads / 64; ++w) {
    const int v = s_scan[w];
    if (w < wave) wave_off += v;
    total += v;
  }
  int top = wave_off + incl - heads;
  // nv == 0: the reference reads its zero-filled scratch and ends up with one cell {point 0}
  const int end = nv > 0 ? total : 1;

  // ---- 4. shuffle position tables.  (Round 5: every table in parallel.  Thread 0 alone used to iterate the generator
  // 512 times and then run the 511-step prefix sum as a chain of dependent LDS round trips -- ~50 us of the kernel's
  // 52-72 us whatever N; the values are the same integers.)
  if (tid < 512) {  // T[i]: the generator applied i times to k0, by the thread that owns entry i
    int k = nv > 0 ? key_cell(key_at(0)) % 256 : 0;
    for (int i = 0; i < tid; ++i) k = (17 * k + 139) % 256;
    s_T[tid] = k;
  }
  __syncthreads();
  const int nA = end < 256 ? end : 256;
  if (tid < 256) {
    int e = 0;
    const int v = s_T[tid];
#pragma unroll 8
    for (int i = 0; i < 256; ++i) e += (i < tid && s_T[i] == v) ? 1 : 0;  // same address across the wave: LDS broadcasts
    s_E[tid] = e;
    s_invB[s_T[256 + tid]] = tid;  // T[256..511] is a permutation of 0..255
  }
  if (tid < 511) {
    const int v = tid - 255;
    int c = 0;
#pragma unroll 8
    for (int i = 0; i < 256; ++i) c += (i < nA && s_T[i] == v) ? 1 : 0;
    s_cntA[tid] = c;
  }
  __syncthreads();
  {  // base[vi] = exclusive prefix sum of (cells with key v among the first 256) + (among the rest), v = vi - 255
    int term = 0;
    if (tid < 511) {
      const int nB = end - 256;  // number of cells with i >= 256 (may be <= 0)
      const int v = tid - 255;
      int cB = 0;
      if (v >= 0 && nB > 0) {
        const int r = s_invB[v];
        if (nB > r) cB = ((nB - 1 - r) >> 8) + 1;
      }
      term = s_cntA[tid] + cB;
    }
    int inc = term;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(inc, off, 64);
      if (lane >= off) inc += v;
    }
    if (lane == 63 && wave < 8) s_scan2[wave] = inc;
    __syncthreads();
    if (tid < 512) {
      int before = 0;
      for (int w = 0; w < wave; ++w) before += s_scan2[w];
      s_base[tid] = before + inc - term;
    }
  }
  __syncthreads();

  auto position = [&](int i) -> int {
    if (i < 256) return s_base[s_T[i] + 255] + s_E[i];
    const int r = (i - 256) & 255;
    const int v = s_T[256 + r];
    return s_base[v + 255] + s_cntA[v + 255] + ((i - 256) >> 8);
  };

  // ---- barycentres, written straight to their shuffled slot
  if (nv == 0) {
    if (tid == 0 && m > 0) {
      o[0] = p[0] / 1.0f;
      o[1] = p[1] / 1.0f;
      o[2] = p[2] / 1.0f;
      om[0] = 1;
    }
  } else {
    for (int pos = r0; pos < r1; ++pos) {
      const int cell = key_cell(key_at(pos));
      if (!(pos == 0 || cell != key_cell(key_at(pos - 1)))) continue;
      int j = key_orig(key_at(pos));
      float xs = p[j * 3 + 0], ys = p[j * 3 + 1], zs = p[j * 3 + 2];
      float pnum = 1.0f;
      for (int pp = pos + 1; pp < nv; ++pp) {
        const unsigned long long kk = key_at(pp);
        if (key_cell(kk) != cell) break;
        j = key_orig(kk);
        xs += p[j * 3 + 0];
        ys += p[j * 3 + 1];
        zs += p[j * 3 + 2];
        pnum += 1.0f;
      }
      const int dst = position(top);
      if (dst < m) {
        o[dst * 3 + 0] = xs / pnum;
        o[dst * 3 + 1] = ys / pnum;
        o[dst * 3 + 2] = zs / pnum;
        om[dst] = 1;
      }
      ++top;
    }
  }
  __syncthreads();

  // ---- 5. wrap-around padding
  for (int i = end + tid; i < m; i += kSubThreads) {
    const int src = i % end;
    o[i * 3 + 0] = o[src * 3 + 0];
    o[i * 3 + 1] = o[src * 3 + 1];
    o[i * 3 + 2] = o[src * 3 + 2];
    om[i] = 0;
  }
}

}  // namespace cl3d

namespace cl3d {
size_t grid_subsampling_workspace(int B, int N) {
  if (N <= kSubMaxN) return 0;
  const size_t n = (size_t)B * N;
  size_t temp = 0;
  (void)rocprim::radix_sort_keys(nullptr, temp, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                 (unsigned)n, 0, 64, (hipStream_t)0);
  return 256 * ((sizeof(SubParams) * B + 255) / 256) + 2 * ((n * 8 + 255) & ~(size_t)255) + temp;
}
}  // namespace cl3d

extern "C" int cl3d_masked_grid_subsampling(const float *xyz, const int32_t *mask, int B, int N,
                                            int m, float sampleDl, float *sub_xyz,
                                            int32_t *sub_mask, void *ws, size_t ws_bytes,
                                            cl3d_stream_t stream) {
  CL3D_REQUIRE(B >= 0 && N >= 1 && m >= 0, "grid_subsampling: bad sizes B=%d N=%d m=%d", B, N, m);
  if (B == 0 || m == 0) return CL3D_OK;
  CL3D_REQUIRE(xyz && mask && sub_xyz && sub_mask, "grid_subsampling: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (N > cl3d::kSubMaxN) {
    // large clouds: keys to HBM, one device-wide radix sort, then the same per-cloud kernel on sorted keys
    CL3D_REQUIRE(B <= 64 && N < (1 << cl3d::kBigIdxBits), "grid_subsampling: large-N path supports B <= 64, N < 2^26");
    const size_t need = cl3d::grid_subsampling_workspace(B, N);
    if (!ws || ws_bytes < need) return cl3d::fail(CL3D_E_WORKSPACE, "grid_subsampling: workspace %zu < %zu", ws_bytes, need);
    const size_t n = (size_t)B * N;
    char *p = static_cast<char *>(ws);
    cl3d::SubParams *params = reinterpret_cast<cl3d::SubParams *>(p);
    p += 256 * ((sizeof(cl3d::SubParams) * B + 255) / 256);
    unsigned long long *keys_in = reinterpret_cast<unsigned long long *>(p);
    p += (n * 8 + 255) & ~(size_t)255;
    unsigned long long *keys_out = reinterpret_cast<unsigned long long *>(p);
    p += (n * 8 + 255) & ~(size_t)255;
    size_t temp_bytes = ws_bytes - (size_t)(p - static_cast<char *>(ws));
    hipLaunchKernelGGL((cl3d::grid_subsample_kernel<false, true>), dim3(B), dim3(cl3d::kSubThreads), 0, st, xyz, mask, N,
                       m, sampleDl, N, sub_xyz, sub_mask, keys_in, params);
    hipError_t e = rocprim::radix_sort_keys(p, temp_bytes, (const unsigned long long *)keys_in, keys_out, (unsigned)n,
                                            0, 64, st);
    if (e != hipSuccess) return cl3d::fail(CL3D_E_LAUNCH, "grid_subsampling: radix sort: %s", hipGetErrorString(e));
    hipLaunchKernelGGL((cl3d::grid_subsample_kernel<true, false>), dim3(B), dim3(cl3d::kSubThreads), 0, st, xyz, mask, N,
                       m, sampleDl, N, sub_xyz, sub_mask, keys_out, params);
    return cl3d::check_launch("cl3d_masked_grid_subsampling(large)");
  }
  int P = 2;
  while (P < N) P <<= 1;
  const size_t lds = (size_t)P * sizeof(unsigned long long);
  static std::atomic<unsigned long long> sort_granted{0};
  int rc_lds = cl3d::lds_opt_in(sort_granted, reinterpret_cast<const void *>(cl3d::grid_subsample_kernel<false, false>),
                                128 * 1024, "grid_subsampling");
  if (rc_lds != CL3D_OK) return rc_lds;
  hipLaunchKernelGGL((cl3d::grid_subsample_kernel<false, false>), dim3(B), dim3(cl3d::kSubThreads), lds, st, xyz, mask, N,
                     m, sampleDl, P, sub_xyz, sub_mask, (unsigned long long *)nullptr, (cl3d::SubParams *)nullptr);
  return cl3d::check_launch("cl3d_masked_grid_subsampling");
}

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
#include <type_traits>

#include "ball_query.h"

#ifndef CL3D_SUB_PHASE
#define CL3D_SUB_PHASE 0   // timing builds only: the kernel returns after phase n (scripts/micro/kernel_variants.py)
#endif
#ifndef CL3D_SUB_SORT
#define CL3D_SUB_SORT 1    // in-LDS sort: 1 = stable radix sort on the cell bits (round 6), 0 = the bitonic network (A/B builds)
#endif

namespace cl3d {

constexpr int kSubThreads = 1024;
constexpr int kSubMaxN = 16384;  // 128 KiB of sort keys; the CU has 160 KiB of LDS

__device__ __forceinline__ int key_cell(unsigned long long k) {
  return (int)(((unsigned)(k >> 32)) ^ 0x80000000u);
}
__device__ __forceinline__ int key_orig(unsigned long long k) { return (int)(unsigned)(k & 0xffffffffull); }

// Large clouds (N > kSubMaxN): the (cell, index) keys do not fit LDS.  A first kernel writes them to HBM as
// {cloud:6 | biased cell:32 | index:26}, grid_sort_kernel orders every cloud's slice (one workgroup per cloud, a stable
// LSD radix sort on the cell bits -- see there; rounds 1-5 called rocPRIM's device-wide radix sort here), and
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

// Bitonic network over P = R x 1024 keys (R = 1: P <= 1024 keys on the first P threads), thread t holding elements
// t R .. t R + R - 1 in REGISTERS.  A compare-exchange at distance j stays inside the thread for j < R, goes through a wave
// shuffle for j < 64 R, and only the 10 (P = 4096) cross-wave stages of the 78 go through LDS and barriers.  Against the
// all-in-LDS network (two 64-bit reads, two conditional writes and a barrier per stage: 38.5 us at N = 4096) this one takes
// 30 us -- by early-exit builds (scripts/micro/grid_subsample_phases.py) ~11 us in the 45 shuffle stages, ~4 in the 23
// in-register ones and the rest in the 20 barriers of 16 waves; not the 8 us the instruction count suggests.  Element e keeps the smaller key of the pair (e, e ^ j) iff ((e & j) == 0) == ((e & k) == 0).  Keys are unique,
// so every correct network yields the same array.  Sorted keys are left in lds[0 .. P).
template <int R>
__device__ __forceinline__ void bitonic_sort_regs(unsigned long long (&key)[R], int P, unsigned long long *lds, int tid) {
  const int e0 = tid * R;
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < R) {
#pragma unroll
        for (int J = R >> 1; J >= 1; J >>= 1) {
          if (j != J) continue;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            if (r & J) continue;
            const bool up = ((e0 + r) & k) == 0;
            const unsigned long long a = key[r], c = key[r | J];
            if ((a > c) == up) {
              key[r] = c;
              key[r | J] = a;
            }
          }
        }
      } else if (j < 64 * R) {
        const int d = j / R;  // partner lane = lane ^ d, same r
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const unsigned long long other = __shfl_xor(key[r], d, 64);
          const int e = e0 + r;
          const bool keep_min = ((e & j) == 0) == ((e & k) == 0);
          const bool other_smaller = other < key[r];
          if (keep_min == other_smaller) key[r] = other;
        }
      } else {
        __syncthreads();  // the previous cross-wave stage's readers are done
        if (e0 < P) {
#pragma unroll
          for (int r = 0; r < R; ++r) lds[e0 + r] = key[r];
        }
        __syncthreads();
        if (e0 < P) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int e = e0 + r;
            const unsigned long long other = lds[e ^ j];
            const bool keep_min = ((e & j) == 0) == ((e & k) == 0);
            const bool other_smaller = other < key[r];
            if (keep_min == other_smaller) key[r] = other;
          }
        }
      }
    }
  }
  __syncthreads();
  if (e0 < P) {
#pragma unroll
    for (int r = 0; r < R; ++r) lds[e0 + r] = key[r];
  }
  __syncthreads();
}

// ---- round 6: the in-LDS sort as a stable LSD radix sort on the CELL bits.  Keys are made in index order, so a stable
// sort by cell alone is the sort by (cell, index) -- and a cloud of N points has at most N occupied cells of a grid with a
// few thousand: 10-13 significant bits = TWO 8-bit passes, against the 78 compare-exchange stages (20 of them through
// barriers) of the bitonic network on the full 64-bit keys: 30 us of the kernel's 48 at N = 4096.  Layout: wave w owns the
// elements [w 64 R, (w+1) 64 R), lane l holds the elements w 64 R + 64 b + l (b = 0 .. R-1) in registers for the whole
// pass, so a batch b is 64 consecutive elements and the passes need ONE key buffer in LDS (everything is read into
// registers before anything is scattered).  Per pass: wave-private digit histograms (LDS atomics), cursors
// (digit-major, wave-minor), and a scatter in which the lanes of a batch that share a digit are found with eight ballots
// -- stable within the batch, batches in order, waves in order.  Only the nv valid keys are sorted (the rest of the
// array is never read).  Same result as any correct sort: the keys are unique.
template <int R>
__device__ __forceinline__ void radix_sort_lds(unsigned long long (&key)[R], int nv, unsigned long long *lds,
                                               unsigned (*hist)[256], unsigned *tot, unsigned *wsum, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int wbase = wave * 64 * R;
  // the cell bits in which any two valid keys differ
  if (tid == 0) wsum[16] = (unsigned)(key[0] >> 32);
  __syncthreads();
  const unsigned first = wsum[16];
  unsigned diff = 0;
#pragma unroll
  for (int b = 0; b < R; ++b)
    if (wbase + 64 * b + lane < nv) diff |= (unsigned)(key[b] >> 32) ^ first;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) diff |= __shfl_xor(diff, o, 64);
  if (lane == 0) wsum[wave] = diff;
  __syncthreads();
  diff = 0;
  for (int w = 0; w < kSubThreads / 64; ++w) diff |= wsum[w];
  const int nbits = diff == 0 ? 0 : 32 - __builtin_clz(diff);
  const int passes = (nbits + 7) / 8;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = 32 + 8 * pass;
    __syncthreads();  // (wsum and the histograms of the previous pass have been read)
    for (int i = tid; i < (kSubThreads / 64) * 256; i += kSubThreads) (&hist[0][0])[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < R; ++b)
      if (wbase + 64 * b + lane < nv) atomicAdd(&hist[wave][(unsigned)(key[b] >> shift) & 255u], 1u);
    __syncthreads();
    if (tid < 256) {
      unsigned run = 0;
      for (int w = 0; w < kSubThreads / 64; ++w) {
        const unsigned c = hist[w][tid];
        hist[w][tid] = run;
        run += c;
      }
      tot[tid] = run;
    }
    __syncthreads();
    {
      const unsigned mine = tid < 256 ? tot[tid] : 0u;
      unsigned incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      if (lane == 63 && wave < 4) wsum[wave] = incl;
      __syncthreads();
      unsigned before = 0;
      for (int w = 0; w < 4; ++w)
        if (w < wave) before += wsum[w];
      const unsigned start = before + incl - mine;
      if (tid < 256)
        for (int w = 0; w < kSubThreads / 64; ++w) hist[w][tid] += start;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < R; ++b) {
      const bool on = wbase + 64 * b + lane < nv;
      const unsigned d = (unsigned)(key[b] >> shift) & 255u;
      unsigned long long same = __ballot(on);
      if (same == 0ull) break;  // (uniform: this wave's remaining batches lie beyond the valid keys)
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
        const unsigned long long m = __ballot((d >> bit) & 1u);
        same &= ((d >> bit) & 1u) ? m : ~m;
      }
      const int rank = prefix_popc(same);
      const int count = (int)__popcll(same);
      unsigned base = 0;
      if (on) base = hist[wave][d];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (on && rank == count - 1) hist[wave][d] = base + (unsigned)count;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (on) lds[base + (unsigned)rank] = key[b];
    }
    __syncthreads();
    if (pass + 1 < passes) {
#pragma unroll
      for (int b = 0; b < R; ++b) {
        const int e = wbase + 64 * b + lane;
        key[b] = e < nv ? lds[e] : ~0ull;
      }
    }
  }
  if (passes == 0) {  // one occupied cell (or no valid key): index order is the sorted order
#pragma unroll
    for (int b = 0; b < R; ++b) {
      const int e = wbase + 64 * b + lane;
      if (e < nv) lds[e] = key[b];
    }
  }
  __syncthreads();
}

// ---- the large-N sort (configs 3 / 5: scenes of 40 960 / 81 920 points).  One 1024-thread workgroup per cloud sorts the
// cloud's nv valid keys -- they are written in index order, so a STABLE sort by the cell field alone is the sort by
// (cell, index) the reference's sort_by_key produces (masked_grid_subsampling_gpu.cu:77); the invalid keys already sit
// behind them (valid points first) and are never read.  LSD radix, 8 bits per pass, only over the cell bits in which the
// cloud's keys differ at all (a scene has a few 10^4 .. 10^6 cells: 2-3 passes):
//   count    wave w owns a contiguous piece of the keys and histograms its digits into its own 256 LDS counters
//   scan     cursor[d][w] = keys with a smaller digit + keys with digit d in the pieces before w's (digit-major, wave-minor)
//   scatter  wave w walks its piece again, 64 keys at a time in order; the lanes of a batch that share a digit are found
//            with eight ballots, a lane's position is the cursor + the number of such lanes below it -- stable within the
//            batch, batches in order, pieces in order.
// Ping-pong between the two key buffers; the result always ends in `b` (a last copy when the pass count is even).
constexpr int kSortAhead = 4;  // key batches (of 64) a wave requests before it consumes the first
__global__ __launch_bounds__(kSubThreads) void grid_sort_kernel(unsigned long long *__restrict__ a, unsigned long long *__restrict__ b,
                                                                const SubParams *__restrict__ params, int N) {
  __shared__ unsigned s_hist[kSubThreads / 64][256];
  __shared__ unsigned s_tot[256];
  __shared__ unsigned s_or[kSubThreads / 64];
  const int cloud = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nv = params[cloud].nv;
  unsigned long long *src = a + (size_t)cloud * N, *dst = b + (size_t)cloud * N;
  if (nv <= 0) return;
  // the cell bits in which any two valid keys differ
  const unsigned first = (unsigned)(src[0] >> kBigIdxBits);
  unsigned diff = 0;
  for (int i = tid; i < nv; i += kSubThreads) diff |= (unsigned)(src[i] >> kBigIdxBits) ^ first;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) diff |= __shfl_xor(diff, o, 64);
  if (lane == 0) s_or[wave] = diff;
  __syncthreads();
  diff = 0;
  for (int w = 0; w < kSubThreads / 64; ++w) diff |= s_or[w];
  const int nbits = diff == 0 ? 0 : 32 - __builtin_clz(diff);
  const int passes = (nbits + 7) / 8;
  int piece = (nv + kSubThreads / 64 - 1) / (kSubThreads / 64);
  piece = (piece + 63) & ~63;
  const int p0 = wave * piece < nv ? wave * piece : nv;
  const int p1 = p0 + piece < nv ? p0 + piece : nv;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = kBigIdxBits + 8 * pass;
    for (int i = tid; i < (kSubThreads / 64) * 256; i += kSubThreads) (&s_hist[0][0])[i] = 0u;
    __syncthreads();
    for (int i0 = p0; i0 < p1; i0 += 64 * kSortAhead) {  // kSortAhead batches requested together: one round trip, not four
      unsigned long long k[kSortAhead];
#pragma unroll
      for (int u = 0; u < kSortAhead; ++u) {
        const int i = i0 + 64 * u + lane;
        k[u] = src[i < p1 ? i : p1 - 1];
      }
#pragma unroll
      for (int u = 0; u < kSortAhead; ++u)
        if (i0 + 64 * u + lane < p1) atomicAdd(&s_hist[wave][(unsigned)(k[u] >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 256) {  // digit tid: totals, and the pieces' counters turned into starts relative to the digit's start
      unsigned run = 0;
      for (int w = 0; w < kSubThreads / 64; ++w) {
        const unsigned c = s_hist[w][tid];
        s_hist[w][tid] = run;
        run += c;
      }
      s_tot[tid] = run;
    }
    __syncthreads();
    {  // exclusive scan of the 256 digit totals (held by the first four waves; every wave walks the same barriers)
      const unsigned mine = tid < 256 ? s_tot[tid] : 0u;
      unsigned incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      if (lane == 63 && wave < 4) s_or[wave] = incl;
      __syncthreads();
      unsigned before = 0;
      for (int w = 0; w < 4; ++w)
        if (w < wave) before += s_or[w];
      const unsigned start = before + incl - mine;
      if (tid < 256)
        for (int w = 0; w < kSubThreads / 64; ++w) s_hist[w][tid] += start;
    }
    __syncthreads();
    for (int j0 = p0; j0 < p1; j0 += 64 * kSortAhead) {
      unsigned long long kk[kSortAhead];
#pragma unroll
      for (int u = 0; u < kSortAhead; ++u) {
        const int i = j0 + 64 * u + lane;
        kk[u] = src[i < p1 ? i : p1 - 1];
      }
#pragma unroll
      for (int u = 0; u < kSortAhead; ++u) {
      const int i0 = j0 + 64 * u;
      if (i0 >= p1) break;  // (uniform)
      const int i = i0 + lane;
      const bool on = i < p1;
      const unsigned long long key = on ? kk[u] : 0ull;
      const unsigned d = (unsigned)(key >> shift) & 255u;
      unsigned long long same = __ballot(on);
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
        const unsigned long long m = __ballot((d >> bit) & 1u);
        same &= ((d >> bit) & 1u) ? m : ~m;
      }
      const int rank = prefix_popc(same);
      const int count = (int)__popcll(same);
      unsigned base = 0;
      if (on) base = s_hist[wave][d];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (on && rank == count - 1) s_hist[wave][d] = base + (unsigned)count;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (on) dst[base + (unsigned)rank] = key;
      }
    }
    __syncthreads();  // (every wave's stores are issued and, by the barrier's release, visible to the workgroup)
    unsigned long long *t = src;
    src = dst;
    dst = t;
  }
  // `src` holds the sorted keys now; they belong in b
  unsigned long long *out = b + (size_t)cloud * N;
  if (src != out)
    for (int i = tid; i < nv; i += kSubThreads) out[i] = src[i];
}

template <bool PRESORTED, bool KEYS_ONLY>
__global__ __launch_bounds__(kSubThreads) void grid_subsample_kernel(
    const float *__restrict__ xyz, const int *__restrict__ mask, int N, int m, float dl, int P,
    float *__restrict__ sub_xyz, int *__restrict__ sub_mask, unsigned long long *__restrict__ gkeys,
    SubParams *__restrict__ params, int stage_xyz) {
  extern __shared__ unsigned long long lds_keys[];  // [P] (in-LDS path only), then the cloud's coordinates [3N] when staged
  __shared__ float s_red[6][kSubThreads / 64];
  __shared__ int s_scan[kSubThreads / 64];
  __shared__ int s_scan2[8];
  __shared__ int s_tmp;
  __shared__ int s_T[512];
  __shared__ int s_E[256];
  __shared__ int s_invB[256];
  __shared__ int s_cntA[511];
  __shared__ int s_base[512];
  __shared__ unsigned s_hist[kSubThreads / 64][256];  // radix sort: wave-private digit counters / cursors
  __shared__ unsigned s_tot[256];
  __shared__ unsigned s_wsum[kSubThreads / 64 + 1];

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const float *p = xyz + (size_t)b * N * 3;
  const int *mk = mask + (size_t)b * N;
  float *o = sub_xyz + (size_t)b * m * 3;
  int *om = sub_mask + (size_t)b * m;

  // stage_xyz (in-LDS path, the keys and 12 N bytes fit): the cloud's coordinates are copied to LDS by the bounding-box
  // sweep; the cell ids and, above all, the per-cell folds -- a chain of dependent loads per cell -- read them there
  float *s_xyz = reinterpret_cast<float *>(lds_keys + P);
  const bool staged = !PRESORTED && !KEYS_ONLY && stage_xyz != 0;
  auto coord = [&](int i, int a) -> float { return staged ? s_xyz[i * 3 + a] : p[i * 3 + a]; };
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
      if (staged) s_xyz[i * 3 + a] = v;
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
  if (CL3D_SUB_PHASE == 1) return;
  const float inv = 1.0f / dl;
  const float ox = floorf(mn[0] * inv) * dl;
  const float oy = floorf(mn[1] * inv) * dl;
  const float oz = floorf(mn[2] * inv) * dl;
  const int NX = (int)floorf((mx[0] - ox) / dl) + 1;
  const int NY = (int)floorf((mx[1] - oy) / dl) + 1;

  // ---- 2. keys + bitonic sort
  auto make_key = [&](int i) -> unsigned long long {
    unsigned long long key = ~0ull;
    if (i < nv) {
      const int iX = (int)floorf((coord(i, 0) - ox) / dl);
      const int iY = (int)floorf((coord(i, 1) - oy) / dl);
      const int iZ = (int)floorf((coord(i, 2) - oz) / dl);
      const int cell = iX + NX * iY + NX * NY * iZ;
      key = ((unsigned long long)(((unsigned)cell) ^ 0x80000000u) << 32) | (unsigned)i;
    }
    return key;
  };
  if constexpr (KEYS_ONLY) {
    for (int i = tid; i < P; i += kSubThreads) {
      const unsigned long long key = make_key(i);
      if (i < N) gkeys[(size_t)b * N + i] = big_key(b, (int)(((unsigned)(key >> 32)) ^ 0x80000000u), i, i < nv);
    }
    if (tid == 0) params[b].nv = nv;
    return;
  } else {
    auto sort_with = [&](auto rtag) {
      constexpr int R = decltype(rtag)::value;
      unsigned long long key[R];
      if (CL3D_SUB_SORT == 1) {  // (round 6) stable radix sort on the cell bits; keys in the wave-batch layout it works on
        const int wbase = wave * 64 * R;
#pragma unroll
        for (int r = 0; r < R; ++r) key[r] = make_key(wbase + 64 * r + lane);  // (~0 beyond the valid points)
        radix_sort_lds<R>(key, nv, lds_keys, s_hist, s_tot, s_wsum, tid);
        return;
      }
#pragma unroll
      for (int r = 0; r < R; ++r) key[r] = tid * R + r < P ? make_key(tid * R + r) : ~0ull;
      if (CL3D_SUB_PHASE != 6) {
        bitonic_sort_regs<R>(key, P, lds_keys, tid);
      } else {  // (timing build 6: keys made, not sorted)
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (tid * R + r < P) lds_keys[tid * R + r] = key[r];
        __syncthreads();
      }
    };
    switch (P / kSubThreads) {  // keys per thread (P is a power of two, <= kSubMaxN = 16 x 1024)
      case 0: case 1: sort_with(std::integral_constant<int, 1>{}); break;
      case 2: sort_with(std::integral_constant<int, 2>{}); break;
      case 4: sort_with(std::integral_constant<int, 4>{}); break;
      case 8: sort_with(std::integral_constant<int, 8>{}); break;
      default: sort_with(std::integral_constant<int, 16>{}); break;
    }
  }

  }  // !PRESORTED
  if (CL3D_SUB_PHASE == 2 || CL3D_SUB_PHASE >= 6) return;

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

  if (CL3D_SUB_PHASE == 3) return;
  // ---- 4. shuffle position tables.  (Round 5: every table in parallel.  Thread 0 alone used to iterate the generator
  // 512 times and then run the 511-step prefix sum as a chain of dependent LDS round trips -- ~50 us of the kernel's
  // 52-72 us whatever N; the values are the same integers.)
  // T[i]: the generator k -> (17 k + 139) % 256 applied i times to k0.  From a non-negative k0 -- every cell id below
  // 2^31 -- values stay non-negative, `%` is `& 255`, and the generator has FULL period (c odd, a - 1 a multiple of 4):
  // T[0..255] is a permutation of 0..255 and T[256 + r] = T[r].  So no key repeats among the first 256 (E = 0), value v
  // is counted there iff its position is below nA, and the inverse of the second block is the position table itself.
  const int k0 = nv > 0 ? key_cell(key_at(0)) % 256 : 0;
  const int nA = end < 256 ? end : 256;
  if (k0 >= 0) {
    if (tid < 512) {
      // i applications of an affine map are ONE affine map (a_i, c_i), composed from the powers of two in i
      unsigned A = 1u, C = 0u, a = 17u, c = 139u;
      for (int bit = tid; bit != 0; bit >>= 1) {
        if (bit & 1) {
          A = (a * A) & 255u;
          C = (a * C + c) & 255u;
        }
        c = (a * c + c) & 255u;
        a = (a * a) & 255u;
      }
      const int k = (int)((A * (unsigned)k0 + C) & 255u);
      s_T[tid] = k;
      if (tid < 256) {
        s_E[tid] = 0;
        s_invB[k] = tid;  // position of value k in either block
      }
    }
    __syncthreads();
    if (tid < 511) {
      const int v = tid - 255;
      s_cntA[tid] = (v >= 0 && s_invB[v] < nA) ? 1 : 0;
    }
  } else {
    // a negative first key (a cell id past 2^31 wrapped): C++'s signed remainder step by step, as the reference does it,
    // and the general counting loops
    if (tid < 512) {
      int k = k0;
      for (int i = 0; i < tid; ++i) k = (17 * k + 139) % 256;
      s_T[tid] = k;
    }
    __syncthreads();
    if (tid < 256) {
      int e = 0;
      const int v = s_T[tid];
      for (int i = 0; i < tid; ++i) e += (s_T[i] == v) ? 1 : 0;
      s_E[tid] = e;
      s_invB[s_T[256 + tid]] = tid;  // T[256..511] is a permutation of 0..255
    }
    if (tid < 511) {
      const int v = tid - 255;
      int c = 0;
      for (int i = 0; i < nA; ++i) c += (s_T[i] == v) ? 1 : 0;
      s_cntA[tid] = c;
    }
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

  if (CL3D_SUB_PHASE == 4) return;
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
      float xs = coord(j, 0), ys = coord(j, 1), zs = coord(j, 2);
      float pnum = 1.0f;
      for (int pp = pos + 1; pp < nv; ++pp) {
        const unsigned long long kk = key_at(pp);
        if (key_cell(kk) != cell) break;
        j = key_orig(kk);
        xs += coord(j, 0);
        ys += coord(j, 1);
        zs += coord(j, 2);
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

  if (CL3D_SUB_PHASE == 5) return;
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
  return 256 * ((sizeof(SubParams) * B + 255) / 256) + 2 * ((n * 8 + 255) & ~(size_t)255);
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
    // large clouds: keys to HBM, the per-cloud radix sort (grid_sort_kernel), then the same per-cloud kernel on sorted keys
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
    hipLaunchKernelGGL((cl3d::grid_subsample_kernel<false, true>), dim3(B), dim3(cl3d::kSubThreads), 0, st, xyz, mask, N,
                       m, sampleDl, N, sub_xyz, sub_mask, keys_in, params, 0);
    hipLaunchKernelGGL(cl3d::grid_sort_kernel, dim3(B), dim3(cl3d::kSubThreads), 0, st, keys_in, keys_out,
                       (const cl3d::SubParams *)params, N);
    hipLaunchKernelGGL((cl3d::grid_subsample_kernel<true, false>), dim3(B), dim3(cl3d::kSubThreads), 0, st, xyz, mask, N,
                       m, sampleDl, N, sub_xyz, sub_mask, keys_out, params, 0);
    return cl3d::check_launch("cl3d_masked_grid_subsampling(large)");
  }
  int P = 2;
  while (P < N) P <<= 1;
  size_t lds = (size_t)P * sizeof(unsigned long long);
  const int stage_xyz = lds + (size_t)N * 12 <= 120 * 1024 ? 1 : 0;  // + ~26 KB of static tables (8 KB of shuffle tables, 17 KB of radix-sort counters): within the CU's 160 KB
  if (stage_xyz) lds += (size_t)N * 12;
  static std::atomic<unsigned long long> sort_granted{0};
  int rc_lds = cl3d::lds_opt_in(sort_granted, reinterpret_cast<const void *>(cl3d::grid_subsample_kernel<false, false>),
                                128 * 1024, "grid_subsampling");
  if (rc_lds != CL3D_OK) return rc_lds;
  hipLaunchKernelGGL((cl3d::grid_subsample_kernel<false, false>), dim3(B), dim3(cl3d::kSubThreads), lds, st, xyz, mask, N,
                     m, sampleDl, P, sub_xyz, sub_mask, (unsigned long long *)nullptr, (cl3d::SubParams *)nullptr, stage_xyz);
  return cl3d::check_launch("cl3d_masked_grid_subsampling");
}

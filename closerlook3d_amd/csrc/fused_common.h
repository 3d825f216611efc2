// fused_common.h -- shared pieces of the fused local-aggregation kernels (gfx950).
//
// Data layout the fused kernels work in (DESIGN.md "fused operators"):
//   * features are POINT-MAJOR rows  ft[b][i][0..C)   (C*4 bytes contiguous per point), so one
//     neighbour = one contiguous row read: a lane group of L lanes x 16 B covers a row with a single
//     global_load_dwordx4 -- coalesced, any N, no LDS staging of whole feature rows;
//   * a wavefront is split into QW = 64/L lane groups; each group owns one query (forward) or one
//     support point (backward) and walks its neighbour slots sequentially, so every reduction over
//     K lives in registers of one lane -- no cross-lane traffic, no atomics;
//   * per-slot scalars (neighbour index, relative position, mask weight) are prepared once per block
//     into LDS (forward) or read from the `slotrec` array the forward pass left behind (backward);
//   * the backward pass is a GATHER too: `inv_off/inv_slots` is the CSR inverse of idx (for every
//     support point the ascending list of (j,k) slots that reference it), so gradients w.r.t. the
//     features are summed in a fixed order without atomics.
#pragma once
#include "cl3d_common.h"
#include <stdlib.h>

namespace cl3d {

struct LaneMap {
  int L;       // lanes per row
  int QW;      // rows per wave (64 / L)
  int chunks;  // passes over the channel axis: chunk c covers channels [(c*L+cl)*V, +V)
};

// How a row of C channels is cut over the lanes of a wave: L lanes x V channels per piece, QW = 64 / L rows side by side,
// `chunks` passes over the channel axis.  Rounds 1-5 minimised chunks / QW (wave-passes per row) -- for 72 channels 9 lanes x
// 7 rows, two passes.  But the gather passes are bound by the cache lines a wave-load touches, not by issue slots: a piece of
// L x 16 B that starts anywhere costs ~L/8 + 1 lines, so 7 rows x 144 B are 15 lines for 63 lanes where 3 rows x 288 B are
// 10 for 54, and every pass repeats the per-slot index / coordinate work.  Measured on the PointWiseMLP step (round 6,
// profiles/r06/session19*_summary.txt; TRAIN + support-major pass, us): 72 channels 18 lanes 156.7 / 9 lanes 180 / 12 lanes
// 201; 144 channels 18 lanes 96.4 / 12 lanes 103.5 / 36 lanes 109; 288 channels 24 lanes 65.4 / 12 lanes 71.1; 36 and 64
// channels: 9 and 16 lanes (one pass) as before.  The rule that reproduces those choices: wave-passes per row times
// (1 + 6 / L), wider pieces preferred unless a narrower cut is 8 % better, and L then shrunk to what the pass count needs
// (to a multiple of 8 lanes = 128 B where that keeps the rows per wave).  `wide` false keeps the old rule: PosPool's sin / cos
// embedding is bound by its arithmetic per (slot, channel), not by the gathers, and wants the fewest wave-passes per row
// (config 5, sin_cos at 144-1152 channels: 19.8 ms with the old maps, 20.4 with the wide ones; session20_summary.txt).
#ifndef CL3D_LANE_RULE
#define CL3D_LANE_RULE 1  // (0: rounds 1-5's rule, the A/B arm of scripts/micro/kernel_variants.py)
#endif
inline LaneMap pick_lane_map(int C, int V, int max_L = 64, bool wide = true) {
  const int rowv = (C + V - 1) / V;
#ifdef CL3D_LANE_ENV  // variant builds of scripts/micro/kernel_variants.py only: CL3D_LANES = lanes per row
  if (const char *e = getenv("CL3D_LANES")) {
    const int L = atoi(e);
    if (L >= 1 && L <= 64) return LaneMap{L, 64 / L, (rowv + L - 1) / L};
  }
#endif
  double best = 1e30;
  LaneMap m{1, 64, rowv};
  for (int L = max_L; L >= 1; --L) {
    const int qw = 64 / L;
    const int chunks = (rowv + L - 1) / L;
    double cost = (double)chunks / qw;
    if (CL3D_LANE_RULE && wide) cost *= 1.0 + 6.0 / (double)(L < rowv ? L : rowv);
    if (cost < best * 0.92) {  // only go narrower for a real gain
      best = cost;
      m = LaneMap{L, qw, chunks};
    }
  }
  if (CL3D_LANE_RULE && wide) {
    const int need = (rowv + m.chunks - 1) / m.chunks, need8 = (need + 7) & ~7;
    if (need8 <= max_L && 64 / need8 == m.QW) m.L = need8;
    else if (64 / need == m.QW) m.L = need;
  }
  return m;
}

// host-side capability queries behind cl3d_fused_supported(): each is defined next to the launch code that
// enforces the same limit (LDS needed by the per-block slot tile grows with nsample)
bool fused_reduce_supported(int op, int K, int C);  // fused_reduce.hip
bool pwmlp_supported(int K, int Co);                // fused_pwmlp.hip
bool maxpool_supported(int K, int C);               // fused_maxpool.hip

// ---- XCD-aware tile order.  The fused kernels gather point-major rows of one cloud over and over
// (C*4 bytes per neighbour); a cloud's rows (1-2 MB) fit the 4 MiB L2 of one XCD, all clouds together
// do not.  The dispatcher is observed to place workgroup b on XCD b % 8 (MI355X_MICROARCH.md), so the
// virtual tile id v = x + 8*s is decoded as "XCD x works through clouds x, x+8, ... one after the
// other".  This is a speed hint only: any placement computes the same result.  Used when B is a
// multiple of 8; otherwise tiles are simply cloud-major.
// grid sizes of grid-stride kernels are kept multiples of 8 so a block stays on its XCD across iterations
inline int round_grid(long long want, int cap) {
  long long g = want < cap ? want : cap;
  if (g >= 8) g &= ~7LL;
  return (int)(g < 1 ? 1 : g);
}
// Channel chunks go over gridDim.y when the tile grid alone would leave the chip under-filled: the deep layers
// of a backbone have few points and many channels (one 160-point cloud x 1152 channels).  Every kernel walks
// `for (ch = blockIdx.y; ch < chunks; ch += gridDim.y)`, so any value in [1, chunks] is correct.
inline int chunk_grid(long long tiles, int chunks) {
  if (tiles >= 2048 || chunks <= 1) return 1;
  const long long want = (2048 + tiles - 1) / (tiles > 0 ? tiles : 1);
  return (int)(want < chunks ? want : chunks);
}
inline int virtual_tiles(int B, int tiles_per_cloud) {
  return B * tiles_per_cloud;
}
__device__ __forceinline__ void decode_tile(int v, int B, int tiles_per_cloud, int &b, int &tile) {
  if ((B & 7) == 0) {
    const int x = v & 7, s = v >> 3;
    b = x + 8 * (s / tiles_per_cloud);
    tile = s - (s / tiles_per_cloud) * tiles_per_cloud;
  } else {
    b = v / tiles_per_cloud;
    tile = v - b * tiles_per_cloud;
  }
}

template <int V>
struct Vec {
  float v[V];
};

template <int V>
__device__ __forceinline__ Vec<V> load_row(const float *p) {
  Vec<V> r;
  if constexpr (V == 4) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = p[i];
  }
  return r;
}

template <int V>
__device__ __forceinline__ void store_row(float *p, const Vec<V> &r) {
  if constexpr (V == 4) {
    *reinterpret_cast<float4 *>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) p[i] = r.v[i];
  }
}

// Walk the K neighbour slots of one query in batches of KB: KB slot records are read from LDS, then KB
// row gathers are issued back to back, and only then consumed -- KB independent global loads in flight
// per lane (the compiler does not build this batching out of a plain unrolled loop).
template <int V, int KB, class F>
__device__ __forceinline__ void for_each_slot(const float4 *myslots, int K, const float *rows, int row_stride,
                                              int c0, F &&f) {
  for (int k0 = 0; k0 < K; k0 += KB) {
    float4 sr[KB];
    Vec<V> gr[KB];
#pragma unroll
    for (int u = 0; u < KB; ++u) sr[u] = myslots[k0 + u < K ? k0 + u : K - 1];
#pragma unroll
    for (int u = 0; u < KB; ++u) gr[u] = load_row<V>(rows + (size_t)__float_as_int(sr[u].x) * row_stride + c0);
#pragma unroll
    for (int u = 0; u < KB; ++u)
      if (k0 + u < K) f(k0 + u, sr[u], gr[u]);
  }
}

}  // namespace cl3d

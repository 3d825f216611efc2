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

namespace cl3d {

struct LaneMap {
  int L;       // lanes per row
  int QW;      // rows per wave (64 / L)
  int chunks;  // passes over the channel axis: chunk c covers channels [(c*L+cl)*V, +V)
};

// choose lanes-per-row to keep as many of the 64 lanes busy as possible, preferring wide rows
inline LaneMap pick_lane_map(int C, int V) {
  const int rowv = (C + V - 1) / V;
  double best = 1e30;
  LaneMap m{1, 64, rowv};
  for (int L = 64; L >= 1; --L) {
    const int qw = 64 / L;
    const int chunks = (rowv + L - 1) / L;
    const double cost = (double)chunks / qw;
    if (cost < best * 0.92) {  // only go narrower for a real gain
      best = cost;
      m = LaneMap{L, qw, chunks};
    }
  }
  return m;
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

}  // namespace cl3d

// ball_query.hip -- masked ordered ball query and masked nearest query for gfx950.
//
// Replaces masked_ordered_ball_query_gpu.cu:11-96 and masked_nearest_query_gpu.cu:8-62 of the
// reference (one thread per query, one block per cloud, serial scan, scratch in global memory,
// in-thread sort).  Design here, MI355X-first:
//
//   * one wavefront owns QW queries; the 64 lanes hold 64 consecutive support points in
//     registers and the QW query positions are wave-uniform (SGPR) operands, so every support
//     load is amortised over QW distance evaluations and no per-lane candidate list exists;
//   * "first 3*nsample in-radius points in support-index order" (the reference's candidate
//     rule) falls out of the lane order: a ballot of the in-radius predicate plus mbcnt gives
//     every hit its position in index order, and the candidates go to a per-query LDS list;
//   * the running strict minimum (needed for the reference's "patch the last slot" step) is
//     tracked per lane and reduced once per query as a 64-bit (d2 bits, index) key;
//   * the reference's stable sort by distance is a wave-parallel rank sort on
//     (d2, candidate position) out of LDS -- same permutation as any stable sort;
//   * grid = (ceil(M / (4*QW)), B) blocks of 4 waves: thousands of blocks instead of B.
//
// Results are bit-identical to the reference semantics (tests/test_native_gpu.py); the
// distance uses cl3d::dist2's canonical operation order.
#include "ball_query.h"
#include <stdlib.h>
#include <string.h>

namespace cl3d {

constexpr int kWavesPerBlock = 4;

template <int QW>
__global__ __launch_bounds__(256) void ball_query_kernel(
    const float *__restrict__ query_xyz, const float *__restrict__ support_xyz,
    const int *__restrict__ query_mask, const int *__restrict__ support_mask, int M, int N,
    float radius2, int K, int *__restrict__ idx, int *__restrict__ idx_mask,
    const int *__restrict__ only_flagged) {
  extern __shared__ int smem[];
  __shared__ int s_nv;
  const int cap = 3 * K;
  const int b = blockIdx.y;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nq_block = kWavesPerBlock * QW;
  if (only_flagged != nullptr) {  // redo pass after the cell-grid search: most workgroups have nothing to do
    int any = 0;
    for (int t = threadIdx.x; t < nq_block; t += 256) {
      const int j = blockIdx.x * nq_block + t;
      if (j < M && only_flagged[(size_t)b * M + j] != 0) any = 1;
    }
    if (!__syncthreads_or(any)) return;
  }

  // LDS carve: candidate distances, candidate indices, sorted indices
  float *cand_d = reinterpret_cast<float *>(smem) + (size_t)wave * QW * cap;
  int *cand_i = smem + (size_t)nq_block * cap + (size_t)wave * QW * cap;
  int *sorted_i = smem + (size_t)2 * nq_block * cap + (size_t)wave * QW * K;

  const float *q = query_xyz + (size_t)b * M * 3;
  const float *s = support_xyz + (size_t)b * N * 3;
  const int nv = block_first_zero(support_mask + (size_t)b * N, N, &s_nv);

  const int j0 = blockIdx.x * nq_block + wave * QW;  // first query of this wave (uniform)

  float qx[QW], qy[QW], qz[QW];
  float lmin[QW];
  int lidx[QW];
  int cnt[QW];
#pragma unroll
  for (int t = 0; t < QW; ++t) {
    int j = j0 + t;
    j = j < M ? j : M - 1;
    qx[t] = q[j * 3 + 0];
    qy[t] = q[j * 3 + 1];
    qz[t] = q[j * 3 + 2];
    lmin[t] = radius2;
    lidx[t] = 0;
    cnt[t] = 0;
  }

  for (int base = 0; base < nv; base += CL3D_WAVE) {
    const int k = base + lane;
    const bool valid = k < nv;
    const int kk = valid ? k : nv - 1;
    const float sx = s[kk * 3 + 0];
    const float sy = s[kk * 3 + 1];
    const float sz = s[kk * 3 + 2];
#pragma unroll
    for (int t = 0; t < QW; ++t) {
      const float d2 = dist2(qx[t], qy[t], qz[t], sx, sy, sz);
      const bool hit = valid && (d2 < radius2);
      const unsigned long long m = __ballot(hit);
      if (m != 0ull) {
        if (hit && d2 < lmin[t]) {
          lmin[t] = d2;
          lidx[t] = k;
        }
        const int c = cnt[t];
        if (c < cap) {
          const int pos = c + prefix_popc(m);
          if (hit && pos < cap) {
            cand_d[t * cap + pos] = d2;
            cand_i[t * cap + pos] = k;
          }
          const int nc = c + (int)__popcll(m);
          cnt[t] = nc < cap ? nc : cap;
        }
      }
    }
  }
  __syncthreads();

  // patch-up: if the candidate list was cut at 3K and the global strict minimum lies beyond
  // its last entry, the last entry is replaced by the minimum (reference :72-75).
#pragma unroll
  for (int t = 0; t < QW; ++t) {
    unsigned long long key = ((unsigned long long)__float_as_uint(lmin[t]) << 32) | (unsigned)lidx[t];
    key = wave_min_u64(key);
    const int c = __builtin_amdgcn_readfirstlane(cnt[t]);
    cnt[t] = c;
    if (cap > 0 && c >= cap) {
      const int gidx = (int)(unsigned)(key & 0xffffffffull);
      const int last = cand_i[t * cap + cap - 1];
      if (gidx > last && lane == 0) {
        cand_i[t * cap + cap - 1] = gidx;
        cand_d[t * cap + cap - 1] = __uint_as_float((unsigned)(key >> 32));
      }
    }
  }
  __syncthreads();

  // stable sort by distance == rank by (d2, list position)
#pragma unroll
  for (int t = 0; t < QW; ++t) {
    const int c = cnt[t];
    const float *cd = cand_d + t * cap;
    for (int e = lane; e < c; e += CL3D_WAVE) {
      const float de = cd[e];
      int rank = 0;
#pragma unroll 8
      for (int f = 0; f < c; ++f) {
        const float df = cd[f];
        rank += (df < de || (df == de && f < e)) ? 1 : 0;
      }
      if (rank < K) sorted_i[t * K + rank] = cand_i[t * cap + e];
    }
  }
  __syncthreads();

  const int *qm = query_mask + (size_t)b * M;
#pragma unroll
  for (int t = 0; t < QW; ++t) {
    const int j = j0 + t;
    if (j >= M) break;
    if (only_flagged != nullptr && only_flagged[(size_t)b * M + j] == 0) continue;
    const int c = cnt[t];
    const int qmk = qm[j];
    int *oi = idx + ((size_t)b * M + j) * K;
    int *om = idx_mask + ((size_t)b * M + j) * K;
    for (int i = lane; i < K; i += CL3D_WAVE) {
      int v = 0, mk = 0;
      if (c > 0) {
        v = sorted_i[t * K + (i < c ? i : i % c)];
        mk = (i < c && qmk != 0) ? 1 : 0;
      }
      oi[i] = v;
      om[i] = mk;
    }
  }
}

template <int QW>
__global__ __launch_bounds__(256) void nearest_query_kernel(
    const float *__restrict__ query_xyz, const float *__restrict__ support_xyz,
    const int *__restrict__ query_mask, const int *__restrict__ support_mask, int M, int N,
    int *__restrict__ idx, int *__restrict__ idx_mask) {
  __shared__ int s_nv;
  const int b = blockIdx.y;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const float *q = query_xyz + (size_t)b * M * 3;
  const float *s = support_xyz + (size_t)b * N * 3;
  const int nv = block_first_zero(support_mask + (size_t)b * N, N, &s_nv);
  const int j0 = (blockIdx.x * kWavesPerBlock + wave) * QW;

  float qx[QW], qy[QW], qz[QW], lmin[QW];
  int lidx[QW];
#pragma unroll
  for (int t = 0; t < QW; ++t) {
    int j = j0 + t;
    j = j < M ? j : M - 1;
    qx[t] = q[j * 3 + 0];
    qy[t] = q[j * 3 + 1];
    qz[t] = q[j * 3 + 2];
    lmin[t] = 100.0f;  // reference: min_dist = 100, min_idx = -1 (masked_nearest_query_gpu.cu:37-38)
    lidx[t] = -1;
  }
  for (int base = 0; base < nv; base += CL3D_WAVE) {
    const int k = base + lane;
    const bool valid = k < nv;
    const int kk = valid ? k : nv - 1;
    const float sx = s[kk * 3 + 0];
    const float sy = s[kk * 3 + 1];
    const float sz = s[kk * 3 + 2];
#pragma unroll
    for (int t = 0; t < QW; ++t) {
      const float d2 = dist2(qx[t], qy[t], qz[t], sx, sy, sz);
      if (valid && d2 < lmin[t]) {
        lmin[t] = d2;
        lidx[t] = k;
      }
    }
  }
  const int *qm = query_mask + (size_t)b * M;
#pragma unroll
  for (int t = 0; t < QW; ++t) {
    // lanes that never saw d2 < 100 keep (100.0f, 0xffffffff): larger than every real key
    unsigned long long key = ((unsigned long long)__float_as_uint(lmin[t]) << 32) | (unsigned)lidx[t];
    key = wave_min_u64(key);
    const int j = j0 + t;
    if (j < M && lane == 0) {
      idx[(size_t)b * M + j] = (int)(unsigned)(key & 0xffffffffull);
      idx_mask[(size_t)b * M + j] = qm[j] == 0 ? 0 : 1;
    }
  }
}

template <int QW>
static int launch_ball_query(const float *q, const float *s, const int *qm, const int *sm, int B,
                             int M, int N, float radius, int K, int *idx, int *idx_mask,
                             const int *only_flagged, hipStream_t st) {
  const int nq_block = kWavesPerBlock * QW;
  const size_t lds = (size_t)nq_block * (2 * 3 * K + K) * sizeof(int);
  dim3 grid(ceil_div(M, nq_block), B);
  hipLaunchKernelGGL(ball_query_kernel<QW>, grid, dim3(256), lds, st, q, s, qm, sm, M, N,
                     radius * radius, K, idx, idx_mask, only_flagged);
  return check_launch("cl3d_masked_ordered_ball_query");
}

int ball_query_exhaustive(const float *query_xyz, const float *support_xyz, const int *query_mask,
                          const int *support_mask, int B, int M, int N, float radius, int K, int *idx,
                          int *idx_mask, const int *only_flagged, hipStream_t st) {
  // LDS per block = 4*QW*7K ints; stay within the 64 KiB a kernel gets without opting in.
  const size_t per_q = (size_t)7 * K * sizeof(int);
  const size_t budget = 64 * 1024;
  if (4 * 8 * per_q <= budget && M >= 64)
    return launch_ball_query<8>(query_xyz, support_xyz, query_mask, support_mask, B, M, N, radius, K, idx, idx_mask, only_flagged, st);
  if (4 * 4 * per_q <= budget && M >= 16)
    return launch_ball_query<4>(query_xyz, support_xyz, query_mask, support_mask, B, M, N, radius, K, idx, idx_mask, only_flagged, st);
  if (4 * 1 * per_q <= budget)
    return launch_ball_query<1>(query_xyz, support_xyz, query_mask, support_mask, B, M, N, radius, K, idx, idx_mask, only_flagged, st);
  return fail(CL3D_E_UNSUPPORTED, "ball_query: nsample=%d needs more than 64 KiB of LDS", K);
}

}  // namespace cl3d

// path: 0 = the library's own choice (by size; CL3D_BQ_PATH pins it), 1 = tile (the cloud resident in one CU's LDS),
// 2 = cells (cell grid through HBM scratch), 3 = exhaustive.  Every path returns the same bits; a path that does not apply
// to the sizes (or needs scratch that was not given) is refused, never silently replaced.
static int ball_query_on_path(int path, const float *query_xyz, const float *support_xyz, const int32_t *query_mask,
                              const int32_t *support_mask, int B, int M, int N, float radius, int nsample, int32_t *idx,
                              int32_t *idx_mask, void *ws, size_t ws_bytes, hipStream_t st) {
  CL3D_REQUIRE(B >= 0 && M >= 0 && N >= 1 && nsample >= 1, "ball_query: bad sizes B=%d M=%d N=%d K=%d", B, M, N, nsample);
  CL3D_REQUIRE(path >= 0 && path <= 3, "ball_query: path must be 0 (auto), 1 (tile), 2 (cells) or 3 (exhaustive)");
  if (B == 0 || M == 0) return CL3D_OK;
  CL3D_REQUIRE(query_xyz && support_xyz && query_mask && support_mask && idx && idx_mask, "ball_query: null pointer");
  CL3D_REQUIRE(B <= 65535, "ball_query: B exceeds grid.y limit");
  if (path == 1) {
    if (!cl3d::ball_query_tile_applicable(M, N, nsample))
      return cl3d::fail(CL3D_E_UNSUPPORTED, "ball_query: the tile path does not take M=%d N=%d K=%d", M, N, nsample);
    return cl3d::ball_query_tile(query_xyz, support_xyz, query_mask, support_mask, B, M, N, radius, nsample, idx, idx_mask, st);
  }
  if (path == 2) {
    if (ws == nullptr || !cl3d::ball_query_cells_applicable(M, N, nsample))
      return cl3d::fail(CL3D_E_UNSUPPORTED, "ball_query: the cells path needs scratch and does not take M=%d N=%d K=%d without it",
                        M, N, nsample);
    return cl3d::ball_query_cells(query_xyz, support_xyz, query_mask, support_mask, B, M, N, radius, nsample, idx, idx_mask, ws,
                                  ws_bytes, st);
  }
  if (path == 3)
    return cl3d::ball_query_exhaustive(query_xyz, support_xyz, query_mask, support_mask, B, M, N, radius, nsample, idx, idx_mask,
                                       nullptr, st);
  // CL3D_BQ_PATH=tile|cells|exhaustive pins one implementation where it applies (A/B timing, tests of the
  // less-travelled paths); every path returns the same bits
  static const int pinned = [] {
    const char *e = getenv("CL3D_BQ_PATH");
    if (e == nullptr) return 0;
    return strcmp(e, "tile") == 0 ? 1 : strcmp(e, "cells") == 0 ? 2 : strcmp(e, "exhaustive") == 0 ? 3 : 0;
  }();
  // clouds whose cell-sorted copy fits one CU's LDS: one launch, no scratch, nothing but coordinates and results in HBM
  if ((pinned == 0 || pinned == 1) && cl3d::ball_query_tile_applicable(M, N, nsample))
    return cl3d::ball_query_tile(query_xyz, support_xyz, query_mask, support_mask, B, M, N, radius, nsample, idx,
                                 idx_mask, st);
  // larger clouds: cell-grid search through HBM scratch when scratch is provided and the problem is large enough to
  // pay for the prep pass; otherwise (and for queries too dense for its LDS lists) the exhaustive scan
  if (pinned != 3 && ws != nullptr && cl3d::ball_query_cells_applicable(M, N, nsample))
    return cl3d::ball_query_cells(query_xyz, support_xyz, query_mask, support_mask, B, M, N, radius, nsample, idx,
                                  idx_mask, ws, ws_bytes, st);
  return cl3d::ball_query_exhaustive(query_xyz, support_xyz, query_mask, support_mask, B, M, N, radius, nsample, idx,
                                     idx_mask, nullptr, st);
}

extern "C" int cl3d_masked_ordered_ball_query(const float *query_xyz, const float *support_xyz,
                                              const int32_t *query_mask,
                                              const int32_t *support_mask, int B, int M, int N,
                                              float radius, int nsample, int32_t *idx,
                                              int32_t *idx_mask, void *ws, size_t ws_bytes,
                                              cl3d_stream_t stream) {
  return ball_query_on_path(0, query_xyz, support_xyz, query_mask, support_mask, B, M, N, radius, nsample, idx, idx_mask, ws,
                            ws_bytes, (hipStream_t)stream);
}

extern "C" int cl3d_masked_ordered_ball_query_path(int path, const float *query_xyz, const float *support_xyz,
                                                   const int32_t *query_mask, const int32_t *support_mask, int B, int M,
                                                   int N, float radius, int nsample, int32_t *idx, int32_t *idx_mask,
                                                   void *ws, size_t ws_bytes, cl3d_stream_t stream) {
  return ball_query_on_path(path, query_xyz, support_xyz, query_mask, support_mask, B, M, N, radius, nsample, idx, idx_mask, ws,
                            ws_bytes, (hipStream_t)stream);
}

// bit p set: path p (1 tile, 2 cells -- given scratch of cl3d_workspace_bytes(CL3D_OP_BALL_QUERY, ...) --, 3 exhaustive) takes
// these sizes.  More than one bit: the caller may time them against each other for ITS point density (pt_utils does,
// once per (sizes, radius) key) -- which of tile / cells is faster depends on how many points fall inside the radius,
// a property of the data, not of the sizes.
extern "C" int cl3d_ball_query_paths(int M, int N, int nsample) {
  if (M < 1 || N < 1 || nsample < 1) return 0;
  int mask = 0;
  if (cl3d::ball_query_tile_applicable(M, N, nsample)) mask |= 1 << 1;
  if (cl3d::ball_query_cells_applicable(M, N, nsample)) mask |= 1 << 2;
  if ((size_t)4 * 1 * 7 * nsample * sizeof(int) <= (size_t)64 * 1024) mask |= 1 << 3;
  return mask;
}

extern "C" int cl3d_masked_nearest_query(const float *query_xyz, const float *support_xyz,
                                         const int32_t *query_mask, const int32_t *support_mask,
                                         int B, int M, int N, int32_t *idx, int32_t *idx_mask,
                                         cl3d_stream_t stream) {
  CL3D_REQUIRE(B >= 0 && M >= 0 && N >= 1, "nearest_query: bad sizes B=%d M=%d N=%d", B, M, N);
  if (B == 0 || M == 0) return CL3D_OK;
  CL3D_REQUIRE(query_xyz && support_xyz && query_mask && support_mask && idx && idx_mask, "nearest_query: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (M >= 64) {
    dim3 grid(cl3d::ceil_div(M, cl3d::kWavesPerBlock * 8), B);
    hipLaunchKernelGGL(cl3d::nearest_query_kernel<8>, grid, dim3(256), 0, st, query_xyz, support_xyz, query_mask, support_mask, M, N, idx, idx_mask);
  } else {
    dim3 grid(cl3d::ceil_div(M, cl3d::kWavesPerBlock * 1), B);
    hipLaunchKernelGGL(cl3d::nearest_query_kernel<1>, grid, dim3(256), 0, st, query_xyz, support_xyz, query_mask, support_mask, M, N, idx, idx_mask);
  }
  return cl3d::check_launch("cl3d_masked_nearest_query");
}

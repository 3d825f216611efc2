// ball_query_lds.hip -- masked ordered ball query for clouds whose cell-sorted copy fits one CU's LDS (gfx950).
//
// Same results, bit for bit, as ball_query.hip / ball_query_cells.hip (and therefore as the reference,
// masked_ordered_ball_query_gpu.cu:11-96).  ball_query_cells.hip spends two launches and ~50 MB of HBM traffic on a
// problem of 19 MB: the cell-sorted support records and a task table are written by a prep kernel and read back by
// the query kernel, every task starts with a chain of dependent global round trips, and a third (flag-filtered)
// launch redoes the rare query whose candidate list overflowed.  For N, M <= 4096 -- every layer of the ModelNet
// and PartNet-sized pyramids, and the metric shape -- the whole search structure of a cloud is 64 KiB, so here
// EVERY workgroup builds it for itself in LDS and nothing but the coordinates and the result crosses HBM:
//
//   one launch, grid (P, B), 1024 threads (16 waves, one workgroup per CU):
//   prep    each thread keeps its <= 4 points in registers: bounding box + valid prefix (one block reduce), cell
//           size h >= radius, both histograms with LDS atomics, both scans, and the counting-sort scatter of the
//           support points into `sorted` (float4 {x,y,z, original index}) -- all P workgroups of a cloud do the same
//           (same inputs, same arithmetic; the order inside a cell differs and does not matter);
//   share   workgroup p owns the cells that hold the p-th share of the queries and stages those queries, grouped by
//           cell, as records + a list of tasks (<= 2 queries of one cell) in LDS;
//   query   waves draw tasks from an LDS ticket (dynamic balance at LDS-atomic cost), rebuild the task's 9 candidate
//           runs from the cell ends, stream the window's candidates out of LDS (no global latency anywhere in the
//           loop), and restate the reference's order-dependent rule order-independently exactly as
//           ball_query_cells.hip does (see its header): <= 3K in-radius candidates -> all of them; more -> the 3K
//           smallest original indices with the strict minimum patched into the last slot; result ranked by
//           (distance, original index).
//   dense   a query with more in-radius candidates than its LDS list holds (6K) is finished in place: the 3K-th
//           smallest original index is found by bisection with counting passes over the window, then one pass
//           collects those 3K candidates and the strict minimum.  Slow (a dozen window passes) and rare; no flag
//           array, no second launch.
//
// Candidate list entries are 32 bits: (original index << 16) | position in `sorted` (both < 4096), so the rank by
// original index compares whole entries and the distance is recomputed from the record when an entry is selected
// (dist2 is a pure function: same operands, same bits).
#include "ball_query.h"

namespace cl3d {

constexpr int kT1Threads = 1024;
constexpr int kT1Waves = kT1Threads / CL3D_WAVE;
constexpr int kT1PT = 4;                          // points per thread
constexpr int kT1MaxPts = kT1Threads * kT1PT;     // 4096
constexpr int kT1MaxCells = 2048;
constexpr int kT1QChunk = 512;                    // query records staged per round
constexpr int kT1QW = 2;                          // queries per task
constexpr int kT1CapMul = 6;                      // candidate list holds kT1CapMul*K entries per query
constexpr int kT1Batch = 3;                       // candidate records in flight per lane

__host__ __device__ inline int tl1_pad4(int x) { return (x + 3) & ~3; }

// dynamic LDS layout (in ints)
struct Tl1Layout {
  int sorted, s_end, q_end, qrec, qcell, tasks, wave0, per_wave, total;
  int candS, cap3S, outS;
};
__host__ __device__ inline Tl1Layout tl1_layout(int N, int K) {
  Tl1Layout l;
  l.candS = tl1_pad4(kT1CapMul * K);
  l.cap3S = tl1_pad4(3 * K);
  l.outS = tl1_pad4(K + 1);
  int o = 0;
  l.sorted = o; o += 4 * tl1_pad4(N);
  l.s_end = o; o += kT1MaxCells;
  l.q_end = o; o += kT1MaxCells;
  l.qrec = o; o += 4 * kT1QChunk;
  l.qcell = o; o += 2 * kT1QChunk;
  l.tasks = o; o += kT1QChunk;
  l.wave0 = o;
  l.per_wave = kT1QW * (l.candS + 2 * l.cap3S + l.outS);
  o += kT1Waves * l.per_wave;
  l.total = o;
  return l;
}

__device__ __forceinline__ int tl1_cell_coord(float x, float o, float inv_h) { return (int)floorf((x - o) * inv_h); }

__device__ __forceinline__ void tl1_wave_sync() {
  // LDS traffic of a wave's private lists is in program order; the fence keeps the compiler from reordering it
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float tl1_uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// The candidate window of a task: <= 9 contiguous runs of `sorted`, laid end to end; w_pe<r> = exclusive prefix of
// run r in the window's index space, w_d<r> = start(run r) - w_pe<r>.  Plain local scalars and a macro on purpose:
// held in a struct or an array, the compiler turns the select chain into "select an offset, load the delta from
// scratch memory" before it promotes the aggregate to registers.
#define TL_WINDOW_POS(p)                                                                                      \
  ((p) + ((p) >= w_pe8 ? w_d8 : (p) >= w_pe7 ? w_d7 : (p) >= w_pe6 ? w_d6 : (p) >= w_pe5 ? w_d5 : (p) >= w_pe4 ? w_d4 \
          : (p) >= w_pe3 ? w_d3 : (p) >= w_pe2 ? w_d2 : (p) >= w_pe1 ? w_d1 : w_d0))

__global__ __launch_bounds__(kT1Threads) void bq_tile1_kernel(const float *__restrict__ query_xyz,
                                                            const float *__restrict__ support_xyz,
                                                            const int *__restrict__ query_mask,
                                                            const int *__restrict__ support_mask, int M, int N,
                                                            float radius, float radius2, int K,
                                                            int *__restrict__ idx, int *__restrict__ idx_mask) {
  extern __shared__ __align__(16) int tl1_lds[];
  __shared__ float s_red[6][kT1Waves];
  __shared__ int s_wave[2][kT1Waves];
  __shared__ int s_nv, s_ticket;
  __shared__ int s_share[4];  // first cell, end cell, first query position, end query position of this workgroup

  const Tl1Layout L = tl1_layout(N, K);
  float4 *sorted = reinterpret_cast<float4 *>(tl1_lds + L.sorted);
  int *s_end = tl1_lds + L.s_end;
  int *q_end = tl1_lds + L.q_end;
  float4 *qrec = reinterpret_cast<float4 *>(tl1_lds + L.qrec);
  int2 *qcell = reinterpret_cast<int2 *>(tl1_lds + L.qcell);
  int *tasks = tl1_lds + L.tasks;

  const int b = blockIdx.y;
  const int part = blockIdx.x, nparts = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float *s = support_xyz + (size_t)b * N * 3;
  const float *q = query_xyz + (size_t)b * M * 3;
  const int *sm = support_mask + (size_t)b * N;
  const int *qm = query_mask + (size_t)b * M;
  const bool same = query_xyz == support_xyz && M == N;

  // ---- (1) this thread's points (kept in registers to the end of the prep), valid prefix, bounding box
  float px[kT1PT], py[kT1PT], pz[kT1PT], qx_[kT1PT], qy_[kT1PT], qz_[kT1PT];
  int mk[kT1PT];
#pragma unroll
  for (int u = 0; u < kT1PT; ++u) {
    const int i = u * kT1Threads + tid;
    const int ic = i < N ? i : N - 1;
    mk[u] = sm[ic];
    px[u] = s[ic * 3 + 0];
    py[u] = s[ic * 3 + 1];
    pz[u] = s[ic * 3 + 2];
  }
  if (!same) {
#pragma unroll
    for (int u = 0; u < kT1PT; ++u) {
      const int i = u * kT1Threads + tid;
      const int ic = i < M ? i : M - 1;
      qx_[u] = q[ic * 3 + 0];
      qy_[u] = q[ic * 3 + 1];
      qz_[u] = q[ic * 3 + 2];
    }
  } else {
#pragma unroll
    for (int u = 0; u < kT1PT; ++u) {
      qx_[u] = px[u];
      qy_[u] = py[u];
      qz_[u] = pz[u];
    }
  }
  if (tid == 0) s_nv = N;
  for (int c = tid; c < kT1MaxCells; c += kT1Threads) {
    s_end[c] = 0;
    q_end[c] = 0;
  }
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  int first0 = N;
#pragma unroll
  for (int u = 0; u < kT1PT; ++u) {
    const int i = u * kT1Threads + tid;
    if (i >= N) continue;
    if (mk[u] == 0) {
      first0 = i < first0 ? i : first0;
      continue;
    }
    mn[0] = px[u] < mn[0] ? px[u] : mn[0]; mx[0] = px[u] > mx[0] ? px[u] : mx[0];
    mn[1] = py[u] < mn[1] ? py[u] : mn[1]; mx[1] = py[u] > mx[1] ? py[u] : mx[1];
    mn[2] = pz[u] < mn[2] ? pz[u] : mn[2]; mx[2] = pz[u] > mx[2] ? pz[u] : mx[2];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float omn = __shfl_xor(mn[a], o, 64), omx = __shfl_xor(mx[a], o, 64);
      mn[a] = omn < mn[a] ? omn : mn[a];
      mx[a] = omx > mx[a] ? omx : mx[a];
    }
    if (lane == 0) {
      s_red[a][wave] = mn[a];
      s_red[3 + a][wave] = mx[a];
    }
  }
  __syncthreads();  // s_nv, the cell arrays and s_red are written
  if (first0 < N) atomicMin(&s_nv, first0);
  __syncthreads();
  const int nv = s_nv;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mn[a] = s_red[a][0];
    mx[a] = s_red[3 + a][0];
    for (int ww = 1; ww < kT1Waves; ++ww) {
      mn[a] = s_red[a][ww] < mn[a] ? s_red[a][ww] : mn[a];
      mx[a] = s_red[3 + a][ww] > mx[a] ? s_red[3 + a][ww] : mx[a];
    }
  }
  // cell size: >= radius with slack, grown until the grid fits (every thread computes the same)
  float h = radius * 1.0002f;
  if (!(h > 0.f)) h = 1.0f;
  int nx = 1, ny = 1, nz = 1;
  bool ok = false;
  if (nv > 0 && mn[0] <= mx[0]) {
    for (int it = 0; it < 64 && !ok; ++it) {
      const float inv = 1.0f / h;
      nx = (int)floorf((mx[0] - mn[0]) * inv) + 1;
      ny = (int)floorf((mx[1] - mn[1]) * inv) + 1;
      nz = (int)floorf((mx[2] - mn[2]) * inv) + 1;
      ok = nx > 0 && ny > 0 && nz > 0 && (long long)nx * ny * nz <= kT1MaxCells;
      if (!ok) h *= 1.3f;
    }
  }
  if (!ok) {  // empty or degenerate (inf/nan) cloud: one cell holding everything
    nx = ny = nz = 1;
    h = 3.0e38f;
    mn[0] = mn[1] = mn[2] = 0.f;
  }
  const float inv_h = 1.0f / h;
  const int ncells = nx * ny * nz;
  // cell of a point: id and the coordinates packed {cx | cy << 16, cz}
  auto cell_of = [&](float x, float y, float z, int &packed_xy, int &cz_out) {
    int cx = tl1_cell_coord(x, mn[0], inv_h), cy = tl1_cell_coord(y, mn[1], inv_h), cz = tl1_cell_coord(z, mn[2], inv_h);
    cx = cx < 0 ? 0 : (cx >= nx ? nx - 1 : cx);
    cy = cy < 0 ? 0 : (cy >= ny ? ny - 1 : cy);
    cz = cz < 0 ? 0 : (cz >= nz ? nz - 1 : cz);
    packed_xy = cx | (cy << 16);
    cz_out = cz;
    return cx + nx * (cy + ny * cz);
  };

  // ---- (2) both histograms
  int cs[kT1PT], cq[kT1PT], cqxy[kT1PT], cqz[kT1PT];
#pragma unroll
  for (int u = 0; u < kT1PT; ++u) {
    const int i = u * kT1Threads + tid;
    int dxy, dz;
    cs[u] = cell_of(px[u], py[u], pz[u], dxy, dz);
    cq[u] = cell_of(qx_[u], qy_[u], qz_[u], cqxy[u], cqz[u]);
    if (i < nv) atomicAdd(&s_end[cs[u]], 1);
    if (i < M) atomicAdd(&q_end[cq[u]], 1);
  }
  __syncthreads();

  // ---- (3) two exclusive scans over the cells with shared barriers.  Thread t owns cells [t*per, (t+1)*per).
  constexpr int kPer = kT1MaxCells / kT1Threads;  // 2
  const int per = (ncells + kT1Threads - 1) / kT1Threads;
  const int t0 = tid * per;
  int cnt_s[kPer], cnt_q[kPer];
  int sum[2] = {0, 0};
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    cnt_s[i] = cnt_q[i] = 0;
    if (i < per && t0 + i < ncells) {
      cnt_s[i] = s_end[t0 + i];
      cnt_q[i] = q_end[t0 + i];
      sum[0] += cnt_s[i];
      sum[1] += cnt_q[i];
    }
  }
  int incl[2] = {sum[0], sum[1]};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl[k], o, 64);
      if (lane >= o) incl[k] += v;
    }
    if (lane == 63) s_wave[k][wave] = incl[k];
  }
  __syncthreads();
  {
    int run[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      int woff = 0;
      for (int ww = 0; ww < kT1Waves; ++ww)
        if (ww < wave) woff += s_wave[k][ww];
      run[k] = woff + incl[k] - sum[k];
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      if (i < per && t0 + i < ncells) {
        s_end[t0 + i] = run[0];  // start of the cell: the scatter's cursor, the cell's end once the scatter is done
        run[0] += cnt_s[i];
        q_end[t0 + i] = run[1];
        run[1] += cnt_q[i];
      }
    }
  }
  __syncthreads();

  // ---- (4) this workgroup's share of the cells: first cell whose query start reaches the share boundary
  if (tid < 2) {
    const int pr = part + tid;
    int lo = 0, hi = ncells;
    if (pr <= 0) hi = 0;
    else if (pr >= nparts) lo = ncells;
    else {
      const int target = (int)((long long)M * pr / nparts);
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (q_end[mid] >= target) hi = mid;
        else lo = mid + 1;
      }
    }
    const int c = pr <= 0 ? 0 : lo;
    s_share[tid] = c;
    s_share[2 + tid] = c < ncells ? q_end[c] : M;
  }
  __syncthreads();
  const int c_lo = s_share[0], c_hi = s_share[1], q_lo = s_share[2], q_hi = s_share[3];

  // ---- (5) scatters: every support point into `sorted`, this share's queries into positions of the share
  int qpos[kT1PT];
#pragma unroll
  for (int u = 0; u < kT1PT; ++u) {
    const int i = u * kT1Threads + tid;
    if (i < nv) {
      const int pos = atomicAdd(&s_end[cs[u]], 1);
      sorted[pos] = make_float4(px[u], py[u], pz[u], __int_as_float(i));
    }
    qpos[u] = -1;
    if (i < M && cq[u] >= c_lo && cq[u] < c_hi) qpos[u] = atomicAdd(&q_end[cq[u]], 1) - q_lo;
  }
  // from here on: s_end[c] = end of cell c in `sorted` (start of cell c+1); q_end[c] = end of cell c in query
  // order for the cells of this share

  // per-wave lists
  int *wbase = tl1_lds + L.wave0 + wave * L.per_wave;
  unsigned *cand = reinterpret_cast<unsigned *>(wbase);                       // [QW][candS]
  float *sel_d = reinterpret_cast<float *>(wbase + kT1QW * L.candS);           // [QW][cap3S]
  int *sel_i = wbase + kT1QW * L.candS + kT1QW * L.cap3S;                      // [QW][cap3S]
  int *out_i = wbase + kT1QW * L.candS + 2 * kT1QW * L.cap3S;                  // [QW][outS]
  const int cap = kT1CapMul * K, cap3 = 3 * K;
  const int lane_ry = lane % 3, lane_rz = (lane / 3) % 3;  // run r = lane < 9: row (y0 + r % 3, z0 + r / 3)

  const int nshare = q_hi - q_lo;
  for (int chunk0 = 0; chunk0 < nshare; chunk0 += kT1QChunk) {
    const int cn = nshare - chunk0 < kT1QChunk ? nshare - chunk0 : kT1QChunk;
    __syncthreads();  // the scatter is complete / the previous round's records have been consumed
#pragma unroll
    for (int u = 0; u < kT1PT; ++u) {
      const int r = qpos[u] - chunk0;
      if (qpos[u] >= 0 && r >= 0 && r < kT1QChunk) {
        qrec[r] = make_float4(qx_[u], qy_[u], qz_[u], __int_as_float(u * kT1Threads + tid));
        qcell[r] = make_int2(cq[u] | (cqz[u] << 16), cqxy[u]);
      }
    }
    if (tid == 0) s_ticket = 0;
    __syncthreads();
    // tasks: pairs of queries of one cell, aligned to even offsets inside the cell
    bool start = false;
    int tk = 0;
    if (tid < cn) {
      const int c = qcell[tid].x & 0xffff;
      const int st = c == c_lo ? q_lo : q_end[c - 1];
      const int o = (q_lo + chunk0 + tid) - st;
      const bool even = (o & 1) == 0;
      start = even || tid == 0;
      const bool two = even && tid + 1 < cn && (qcell[tid + 1].x & 0xffff) == c;
      tk = tid | ((two ? 2 : 1) << 16);
    }
    const unsigned long long sm_ = __ballot(start);
    if (lane == 0) s_wave[0][wave] = (int)__popcll(sm_);
    __syncthreads();
    int woff = 0, ntasks = 0;
    for (int ww = 0; ww < kT1Waves; ++ww) {
      const int v = s_wave[0][ww];
      if (ww < wave) woff += v;
      ntasks += v;
    }
    if (start) tasks[woff + prefix_popc(sm_)] = tk;
    __syncthreads();

    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&s_ticket, 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= ntasks) break;
      const int tkv = __builtin_amdgcn_readfirstlane(tasks[t]);
      const int r0 = tkv & 0xffff, n = tkv >> 16;
      int jq[kT1QW];
      float qx[kT1QW], qy[kT1QW], qz[kT1QW];
      int cnt[kT1QW];
#pragma unroll
      for (int u = 0; u < kT1QW; ++u) {
        const float4 qq = qrec[r0 + (u < n ? u : 0)];
        jq[u] = __builtin_amdgcn_readfirstlane(__float_as_int(qq.w));
        // an unused query slot of the task gets a NaN coordinate: its distances are NaN and never "in radius"
        qx[u] = u < n ? tl1_uniform(qq.x) : __builtin_nanf("");
        qy[u] = tl1_uniform(qq.y);
        qz[u] = tl1_uniform(qq.z);
        cnt[u] = 0;
      }
      // the cell's candidate window: 3x3 (y,z) rows of <= 3 x-adjacent cells, each one contiguous run of `sorted`
      int w_pe1, w_pe2, w_pe3, w_pe4, w_pe5, w_pe6, w_pe7, w_pe8, w_d0, w_d1, w_d2, w_d3, w_d4, w_d5, w_d6, w_d7, w_d8, T;
      {
        const int2 cc = qcell[r0];
        const int cz = __builtin_amdgcn_readfirstlane(cc.x >> 16);
        const int cx = __builtin_amdgcn_readfirstlane(cc.y & 0xffff), cy = __builtin_amdgcn_readfirstlane(cc.y >> 16);
        const int x0 = cx > 0 ? cx - 1 : 0, x1 = cx + 1 < nx ? cx + 1 : nx - 1;
        const int y0 = cy > 0 ? cy - 1 : 0, y1 = cy + 1 < ny ? cy + 1 : ny - 1;
        const int z0 = cz > 0 ? cz - 1 : 0, z1 = cz + 1 < nz ? cz + 1 : nz - 1;
        const int yy = y0 + lane_ry, zz = z0 + lane_rz;
        int ra = 0, len = 0;
        if (lane < 9 && yy <= y1 && zz <= z1) {
          const int row = nx * (yy + ny * zz);
          const int ca = row + x0, cb = row + x1 + 1;  // cells [ca, cb): starts are the previous cell's end
          ra = ca > 0 ? s_end[ca - 1] : 0;
          len = s_end[cb - 1] - ra;
        }
        int inc = len;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          const int v = __shfl_up(inc, o, 64);
          if (lane >= o) inc += v;
        }
        const int pe = inc - len, dl = ra - pe;
        w_pe1 = __builtin_amdgcn_readlane(pe, 1); w_pe2 = __builtin_amdgcn_readlane(pe, 2);
        w_pe3 = __builtin_amdgcn_readlane(pe, 3); w_pe4 = __builtin_amdgcn_readlane(pe, 4);
        w_pe5 = __builtin_amdgcn_readlane(pe, 5); w_pe6 = __builtin_amdgcn_readlane(pe, 6);
        w_pe7 = __builtin_amdgcn_readlane(pe, 7); w_pe8 = __builtin_amdgcn_readlane(pe, 8);
        w_d0 = __builtin_amdgcn_readlane(dl, 0); w_d1 = __builtin_amdgcn_readlane(dl, 1);
        w_d2 = __builtin_amdgcn_readlane(dl, 2); w_d3 = __builtin_amdgcn_readlane(dl, 3);
        w_d4 = __builtin_amdgcn_readlane(dl, 4); w_d5 = __builtin_amdgcn_readlane(dl, 5);
        w_d6 = __builtin_amdgcn_readlane(dl, 6); w_d7 = __builtin_amdgcn_readlane(dl, 7);
        w_d8 = __builtin_amdgcn_readlane(dl, 8);
        T = __builtin_amdgcn_readlane(inc, 8);
      }
      // ---- candidates: every in-radius candidate of the window goes to the LDS list (unordered: the selection
      // below restates the reference's order-dependent rule), one ballot + prefix count per query and 64 candidates
      for (int p0 = 0; p0 < T; p0 += CL3D_WAVE * kT1Batch) {
        float4 sp[kT1Batch];
        int sp_pos[kT1Batch];
#pragma unroll
        for (int v = 0; v < kT1Batch; ++v) {
          int p = p0 + v * CL3D_WAVE + lane;
          p = p < T ? p : T - 1;
          sp_pos[v] = TL_WINDOW_POS(p);
          sp[v] = sorted[sp_pos[v]];
        }
#pragma unroll
        for (int v = 0; v < kT1Batch; ++v) {
          if (p0 + v * CL3D_WAVE >= T) break;  // uniform
          const bool live = p0 + v * CL3D_WAVE + lane < T;
          const unsigned entry = ((unsigned)__float_as_int(sp[v].w) << 16) | (unsigned)sp_pos[v];
#pragma unroll
          for (int u = 0; u < kT1QW; ++u) {
            const float d2 = dist2(qx[u], qy[u], qz[u], sp[v].x, sp[v].y, sp[v].z);
            const bool hit = live && (d2 < radius2);
            const unsigned long long m = __ballot(hit);
            const int c0 = cnt[u];
            const int c1 = c0 + (int)__popcll(m);  // wave-uniform
            if (c1 <= cap && hit) cand[u * L.candS + c0 + prefix_popc(m)] = entry;  // an overflowing list is abandoned
            cnt[u] = c1;
          }
        }
      }
      tl1_wave_sync();

#pragma unroll
      for (int u = 0; u < kT1QW; ++u) {
        if (u >= n) continue;
        const int j = jq[u];
        const int S = __builtin_amdgcn_readfirstlane(cnt[u]);
        int *oi = idx + ((size_t)b * M + j) * K;
        int *om = idx_mask + ((size_t)b * M + j) * K;
        unsigned *lc = cand + u * L.candS;
        float *ld = sel_d + u * L.cap3S;
        int *li = sel_i + u * L.cap3S;
        int c = S;
        if (S <= cap3) {
          for (int e = lane; e < S; e += CL3D_WAVE) {
            const unsigned en = lc[e];
            const float4 rec = sorted[en & 0xffffu];
            ld[e] = dist2(qx[u], qy[u], qz[u], rec.x, rec.y, rec.z);
            li[e] = (int)(en >> 16);
          }
        } else {
          // first-occurrence strict minimum == smallest (d2, original index) of all S in-radius candidates
          unsigned long long key = ~0ull;
          int nlist = S;
          if (S > cap) {
            // dense: the list was abandoned.  Bisection for T* = (3K-th smallest original index of S) + 1:
            // count(orig < lo) < 3K <= count(orig < hi)
            int lo = 0, hi = N;
            while (hi - lo > 1) {
              const int mid = (lo + hi) >> 1;
              int below = 0;
              for (int p0 = 0; p0 < T; p0 += CL3D_WAVE) {
                const int p = p0 + lane;
                const float4 rec = sorted[TL_WINDOW_POS(p < T ? p : T - 1)];
                const float d2 = dist2(qx[u], qy[u], qz[u], rec.x, rec.y, rec.z);
                const bool hit = p < T && d2 < radius2 && __float_as_int(rec.w) < mid;
                below += (int)__popcll(__ballot(hit));
              }
              if (below >= cap3) hi = mid;
              else lo = mid;
            }
            int fill = 0;
            for (int p0 = 0; p0 < T; p0 += CL3D_WAVE) {
              const int p = p0 + lane;
              const int pos = TL_WINDOW_POS(p < T ? p : T - 1);
              const float4 rec = sorted[pos];
              const float d2 = dist2(qx[u], qy[u], qz[u], rec.x, rec.y, rec.z);
              const bool hit = p < T && d2 < radius2;
              const int orig = __float_as_int(rec.w);
              if (hit) {
                const unsigned long long ke = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)orig;
                key = ke < key ? ke : key;
              }
              const bool take = hit && orig < hi;
              const unsigned long long m = __ballot(take);
              if (take) lc[fill + prefix_popc(m)] = ((unsigned)orig << 16) | (unsigned)pos;
              fill += (int)__popcll(m);
            }
            nlist = cap3;  // == fill: original indices are distinct
            tl1_wave_sync();
          } else {
            for (int e = lane; e < S; e += CL3D_WAVE) {
              const unsigned en = lc[e];
              const float4 rec = sorted[en & 0xffffu];
              const float d2 = dist2(qx[u], qy[u], qz[u], rec.x, rec.y, rec.z);
              const unsigned long long ke = ((unsigned long long)__float_as_uint(d2) << 32) | (en >> 16);
              key = ke < key ? ke : key;
            }
          }
          key = wave_min_u64(key);
          const int gidx = (int)(unsigned)(key & 0xffffffffull);
          // the 3K smallest original indices, written in index order (an entry orders like its original index)
          const int n4 = nlist & ~3;
          for (int e0 = 0; e0 < nlist; e0 += CL3D_WAVE) {
            const int e = e0 + lane;
            const bool on = e < nlist;
            const unsigned my = on ? lc[e] : 0u;
            int r = 0;
            for (int f = 0; f < n4; f += 4) {
              const uint4 k4 = *reinterpret_cast<const uint4 *>(lc + f);
              r += k4.x < my ? 1 : 0;
              r += k4.y < my ? 1 : 0;
              r += k4.z < my ? 1 : 0;
              r += k4.w < my ? 1 : 0;
            }
            for (int f = n4; f < nlist; ++f) r += lc[f] < my ? 1 : 0;
            if (on && r < cap3) {
              const float4 rec = sorted[my & 0xffffu];
              ld[r] = dist2(qx[u], qy[u], qz[u], rec.x, rec.y, rec.z);
              li[r] = (int)(my >> 16);
            }
          }
          tl1_wave_sync();
          if (gidx > li[cap3 - 1]) {  // uniform: the minimum was cut off -> it takes the last slot
            if (lane == 0) {
              li[cap3 - 1] = gidx;
              ld[cap3 - 1] = __uint_as_float((unsigned)(key >> 32));
            }
          }
          c = cap3;
        }
        // ---- rank by (distance, original index) == stable sort by distance of the index-ordered list.
        // Fast path: rank by the distance alone (d2 >= 0, so its bit pattern orders like the value), keys broadcast
        // four at a time from LDS.  Equal distances give equal ranks and leave a hole in ranks [0, min(c, K+1));
        // a hole is detected below and the exact ranking redoes the (rare) list.
        int *so = out_i + u * L.outS;
        const int need = c < K + 1 ? c : K + 1;
        for (int i = lane; i < need; i += CL3D_WAVE) so[i] = -1;
        tl1_wave_sync();
        const unsigned *lb = reinterpret_cast<const unsigned *>(ld);
        const int c4 = c & ~3;
        for (int e0 = 0; e0 < c; e0 += CL3D_WAVE) {
          const int e = e0 + lane;
          const bool on = e < c;
          const unsigned my = on ? lb[e] : 0u;
          int rank = 0;
          for (int f = 0; f < c4; f += 4) {
            const uint4 k4 = *reinterpret_cast<const uint4 *>(lb + f);
            rank += k4.x < my ? 1 : 0;
            rank += k4.y < my ? 1 : 0;
            rank += k4.z < my ? 1 : 0;
            rank += k4.w < my ? 1 : 0;
          }
          for (int f = c4; f < c; ++f) rank += lb[f] < my ? 1 : 0;
          if (on && rank <= K) so[rank] = li[e];
        }
        tl1_wave_sync();
        bool hole = false;
        for (int i = lane; i < need; i += CL3D_WAVE) hole = hole || so[i] < 0;
        if (__ballot(hole) != 0ull) {  // uniform: ties in distance -> exact (distance, original index) ranking
          for (int e = lane; e < c; e += CL3D_WAVE) {
            const float de = ld[e];
            const int ie = li[e];
            int rank = 0;
#pragma unroll 8
            for (int f = 0; f < c; ++f) {
              const float df = ld[f];
              rank += (df < de || (df == de && li[f] < ie)) ? 1 : 0;
            }
            if (rank <= K) so[rank] = ie;
          }
          tl1_wave_sync();
        }
        const int qmk = qm[j];
        if (c >= K) {  // uniform, the common case: a full list, no wrap-around padding (and no integer modulo)
          for (int i = lane; i < K; i += CL3D_WAVE) {
            oi[i] = so[i];
            om[i] = qmk != 0 ? 1 : 0;
          }
        } else {
          for (int i = lane; i < K; i += CL3D_WAVE) {
            int v = 0, mkv = 0;
            if (c > 0) {
              v = so[i < c ? i : i % c];
              mkv = (i < c && qmk != 0) ? 1 : 0;
            }
            oi[i] = v;
            om[i] = mkv;
          }
        }
      }
      tl1_wave_sync();
    }
  }
}

bool ball_query_tile1_applicable(int M, int N, int K) {
  if (N < 512 || M < 64 || N > kT1MaxPts || M > kT1MaxPts || K < 1) return false;
  return (size_t)tl1_layout(N, K).total * sizeof(int) <= 158 * 1024;
}

int ball_query_tile1(const float *query_xyz, const float *support_xyz, const int *query_mask,
                    const int *support_mask, int B, int M, int N, float radius, int K, int *idx, int *idx_mask,
                    hipStream_t st) {
  if (B > 65535) return fail(CL3D_E_UNSUPPORTED, "ball_query: B exceeds grid.y limit");
  const size_t lds = (size_t)tl1_layout(N, K).total * sizeof(int);
  static std::atomic<unsigned long long> granted{0};
  int rc = lds_opt_in(granted, reinterpret_cast<const void *>(bq_tile1_kernel), 158 * 1024, "ball_query");
  if (rc != CL3D_OK) return rc;
  // workgroups per cloud: one per CU over the whole batch, each with at least 64 queries
  int parts = 256 / (B > 0 ? B : 1);
  const int most = M / 64;
  parts = parts > most ? most : parts;
  parts = parts < 1 ? 1 : (parts > 64 ? 64 : parts);
  hipLaunchKernelGGL(bq_tile1_kernel, dim3(parts, B), dim3(kT1Threads), lds, st, query_xyz, support_xyz, query_mask,
                     support_mask, M, N, radius, radius * radius, K, idx, idx_mask);
  return check_launch("cl3d_masked_ordered_ball_query(tile)");
}

}  // namespace cl3d

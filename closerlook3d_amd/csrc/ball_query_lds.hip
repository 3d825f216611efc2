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
//           cell, as records + a list of tasks (<= kTlQT queries of one cell) in LDS;
//   query   waves draw tasks from an LDS ticket (dynamic balance at LDS-atomic cost).  A task's candidate window -- the
//           <= 9 runs of `sorted` around its cell -- is read out of LDS ONCE into registers (<= 384 records, 24 VGPRs)
//           and serves all the task's queries, two at a time: the two distances of a candidate are one chain of packed
//           FP32 instructions (v_pk_add / v_pk_mul / v_pk_fma: per-lane IEEE, the same bits as the scalar chain of
//           cl3d::dist2), and every in-radius candidate leaves (d2, original index) in the query's LDS list.  The
//           reference's order-dependent rule is then restated order-independently exactly as ball_query_cells.hip
//           does (see its header): <= 3K in-radius candidates -> all of them; more -> the 3K smallest original indices
//           with the strict minimum patched into the last slot; result ranked by (distance, original index).
//   dense   a query with more in-radius candidates than its LDS list holds (6K) is finished in place: the 3K-th
//           smallest original index is found by bisection with counting passes over the window, then one pass
//           collects those 3K candidates and the strict minimum.  Slow (a dozen window passes) and rare; no flag
//           array, no second launch.
//
// Round 4 (this form): window in registers across the queries of a cell, packed distance arithmetic, distances stored
// by the candidate pass instead of being recomputed at selection -- 375 -> ~215 vector instructions per query.
#include "ball_query.h"

namespace cl3d {

constexpr int kTlThreads = 1024;
constexpr int kTlWaves = kTlThreads / CL3D_WAVE;
constexpr int kTlPT = 4;                          // points per thread
constexpr int kTlMaxPts = kTlThreads * kTlPT;     // 4096
constexpr int kTlMaxCells = 2048;
constexpr int kTlQChunk = 512;                    // query records staged per round
constexpr int kTlQT = 4;                          // queries per task (of one cell): they share one window read
constexpr int kTlQW = 2;                          // queries tested per candidate pass (one packed-FP32 chain)
constexpr int kTlCapMul = 6;                      // candidate list holds kTlCapMul*K entries per query
constexpr int kTlWin = 6;                         // window batches (of 64 records) a wave holds in registers
constexpr int kTlIdxRounds = 4;                   // rank-by-index rounds held in registers: kTlCapMul*K <= 256

typedef float tl_v2f __attribute__((ext_vector_type(2)));

__host__ __device__ inline int tl_pad4(int x) { return (x + 3) & ~3; }

// dynamic LDS layout (in ints)
struct TlLayout {
  int sorted, s_end, q_end, qrec, qcell, tasks, wave0, per_wave, total;
  int capS, outS;
};
__host__ __device__ inline TlLayout tl_layout(int N, int K) {
  TlLayout l;
  l.capS = tl_pad4(kTlCapMul * K) + 4;  // + the four sentinel keys the rank loop's 16-byte reads run into
  l.outS = tl_pad4(K + 1);
  int o = 0;
  l.sorted = o; o += 4 * tl_pad4(N);
  l.s_end = o; o += kTlMaxCells;
  l.q_end = o; o += kTlMaxCells;
  l.qrec = o; o += 4 * kTlQChunk;
  l.qcell = o; o += 2 * kTlQChunk;
  l.tasks = o; o += kTlQChunk;
  l.wave0 = o;
  l.per_wave = kTlQW * (2 * l.capS + l.outS);
  o += kTlWaves * l.per_wave;
  l.total = o;
  return l;
}

__device__ __forceinline__ int tl_cell_coord(float x, float o, float inv_h) { return (int)floorf((x - o) * inv_h); }

__device__ __forceinline__ void tl_wave_sync() {
  // LDS traffic of a wave's private lists is in program order; the fence keeps the compiler from reordering it
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float tl_uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// cl3d::dist2 for two queries against one candidate, as packed FP32 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 are
// IEEE per element: each half is bit for bit the scalar chain of dist2; the library is built with -ffp-contract=off,
// so nothing is fused or split behind this).  Same CL3D_D2_FORM switch as cl3d_common.h.
__device__ __forceinline__ tl_v2f tl_dist2_pair(tl_v2f qx, tl_v2f qy, tl_v2f qz, float x, float y, float z) {
  const tl_v2f dx = qx - x, dy = qy - y, dz = qz - z;
#if CL3D_D2_FORM == 0
  const tl_v2f xx = dx * dx;
  const tl_v2f zz = dz * dz;
  return __builtin_elementwise_fma(dy, dy, xx) + zz;
#elif CL3D_D2_FORM == 1
  const tl_v2f xx = dx * dx;
  const tl_v2f yy = dy * dy;
  const tl_v2f zz = dz * dz;
  return (xx + yy) + zz;
#else
  const tl_v2f xx = dx * dx;
  return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, xx));
#endif
}

// The candidate window of a task: <= 9 contiguous runs of `sorted`, laid end to end; w_pe<r> = exclusive prefix of
// run r in the window's index space, w_d<r> = start(run r) - w_pe<r>.  Plain local scalars and a macro on purpose:
// held in a struct or an array, the compiler turns the select chain into "select an offset, load the delta from
// scratch memory" before it promotes the aggregate to registers.
#define TL_WINDOW_POS(p)                                                                                      \
  ((p) + ((p) >= w_pe8 ? w_d8 : (p) >= w_pe7 ? w_d7 : (p) >= w_pe6 ? w_d6 : (p) >= w_pe5 ? w_d5 : (p) >= w_pe4 ? w_d4 \
          : (p) >= w_pe3 ? w_d3 : (p) >= w_pe2 ? w_d2 : (p) >= w_pe1 ? w_d1 : w_d0))

__global__ __launch_bounds__(kTlThreads) void bq_tile_kernel(const float *__restrict__ query_xyz,
                                                            const float *__restrict__ support_xyz,
                                                            const int *__restrict__ query_mask,
                                                            const int *__restrict__ support_mask, int M, int N,
                                                            float radius, float radius2, int K,
                                                            int *__restrict__ idx, int *__restrict__ idx_mask) {
  extern __shared__ __align__(16) int tl_lds[];
  __shared__ float s_red[6][kTlWaves];
  __shared__ int s_wave[2][kTlWaves];
  __shared__ int s_nv, s_ticket;
  __shared__ int s_share[4];  // first cell, end cell, first query position, end query position of this workgroup

  const TlLayout L = tl_layout(N, K);
  float4 *sorted = reinterpret_cast<float4 *>(tl_lds + L.sorted);
  int *s_end = tl_lds + L.s_end;
  int *q_end = tl_lds + L.q_end;
  float4 *qrec = reinterpret_cast<float4 *>(tl_lds + L.qrec);
  int2 *qcell = reinterpret_cast<int2 *>(tl_lds + L.qcell);
  int *tasks = tl_lds + L.tasks;

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
  float px[kTlPT], py[kTlPT], pz[kTlPT], qx_[kTlPT], qy_[kTlPT], qz_[kTlPT];
  int mk[kTlPT];
#pragma unroll
  for (int u = 0; u < kTlPT; ++u) {
    const int i = u * kTlThreads + tid;
    const int ic = i < N ? i : N - 1;
    mk[u] = sm[ic];
    px[u] = s[ic * 3 + 0];
    py[u] = s[ic * 3 + 1];
    pz[u] = s[ic * 3 + 2];
  }
  if (!same) {
#pragma unroll
    for (int u = 0; u < kTlPT; ++u) {
      const int i = u * kTlThreads + tid;
      const int ic = i < M ? i : M - 1;
      qx_[u] = q[ic * 3 + 0];
      qy_[u] = q[ic * 3 + 1];
      qz_[u] = q[ic * 3 + 2];
    }
  } else {
#pragma unroll
    for (int u = 0; u < kTlPT; ++u) {
      qx_[u] = px[u];
      qy_[u] = py[u];
      qz_[u] = pz[u];
    }
  }
  if (tid == 0) s_nv = N;
  for (int c = tid; c < kTlMaxCells; c += kTlThreads) {
    s_end[c] = 0;
    q_end[c] = 0;
  }
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  int first0 = N;
#pragma unroll
  for (int u = 0; u < kTlPT; ++u) {
    const int i = u * kTlThreads + tid;
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
  const int nv = __builtin_amdgcn_readfirstlane(s_nv);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mn[a] = s_red[a][0];
    mx[a] = s_red[3 + a][0];
    for (int ww = 1; ww < kTlWaves; ++ww) {
      mn[a] = s_red[a][ww] < mn[a] ? s_red[a][ww] : mn[a];
      mx[a] = s_red[3 + a][ww] > mx[a] ? s_red[3 + a][ww] : mx[a];
    }
    mn[a] = tl_uniform(mn[a]);  // (the same value in every lane: keep it on the scalar side)
    mx[a] = tl_uniform(mx[a]);
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
      ok = nx > 0 && ny > 0 && nz > 0 && (long long)nx * ny * nz <= kTlMaxCells;
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
    int cx = tl_cell_coord(x, mn[0], inv_h), cy = tl_cell_coord(y, mn[1], inv_h), cz = tl_cell_coord(z, mn[2], inv_h);
    cx = cx < 0 ? 0 : (cx >= nx ? nx - 1 : cx);
    cy = cy < 0 ? 0 : (cy >= ny ? ny - 1 : cy);
    cz = cz < 0 ? 0 : (cz >= nz ? nz - 1 : cz);
    packed_xy = cx | (cy << 16);
    cz_out = cz;
    return cx + nx * (cy + ny * cz);
  };

  // ---- (2) both histograms
  int cs[kTlPT], cq[kTlPT], cqxy[kTlPT], cqz[kTlPT];
#pragma unroll
  for (int u = 0; u < kTlPT; ++u) {
    const int i = u * kTlThreads + tid;
    int dxy, dz;
    cs[u] = cell_of(px[u], py[u], pz[u], dxy, dz);
    cq[u] = cell_of(qx_[u], qy_[u], qz_[u], cqxy[u], cqz[u]);
    if (i < nv) atomicAdd(&s_end[cs[u]], 1);
    if (i < M) atomicAdd(&q_end[cq[u]], 1);
  }
  __syncthreads();

  // ---- (3) two exclusive scans over the cells with shared barriers.  Thread t owns cells [t*per, (t+1)*per).
  constexpr int kPer = kTlMaxCells / kTlThreads;  // 2
  const int per = (ncells + kTlThreads - 1) / kTlThreads;
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
      for (int ww = 0; ww < kTlWaves; ++ww)
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
  const int c_lo = __builtin_amdgcn_readfirstlane(s_share[0]), c_hi = __builtin_amdgcn_readfirstlane(s_share[1]);
  const int q_lo = __builtin_amdgcn_readfirstlane(s_share[2]), q_hi = __builtin_amdgcn_readfirstlane(s_share[3]);

  // ---- (5) scatters: every support point into `sorted`, this share's queries into positions of the share; the
  // share's first kTlQChunk query records are staged straight from the registers (later rounds -- more than 512
  // queries per workgroup, B > 32 -- read the coordinates again: the registers go to the window below)
  int qpos[kTlPT];
#pragma unroll
  for (int u = 0; u < kTlPT; ++u) {
    const int i = u * kTlThreads + tid;
    if (i < nv) {
      const int pos = atomicAdd(&s_end[cs[u]], 1);
      sorted[pos] = make_float4(px[u], py[u], pz[u], __int_as_float(i));
    }
    qpos[u] = -1;
    if (i < M && cq[u] >= c_lo && cq[u] < c_hi) {
      qpos[u] = atomicAdd(&q_end[cq[u]], 1) - q_lo;
      if (qpos[u] < kTlQChunk) {
        qrec[qpos[u]] = make_float4(qx_[u], qy_[u], qz_[u], __int_as_float(i));
        qcell[qpos[u]] = make_int2(cq[u] | (cqz[u] << 16), cqxy[u]);
      }
    }
  }
  // from here on: s_end[c] = end of cell c in `sorted` (start of cell c+1); q_end[c] = end of cell c in query
  // order for the cells of this share

  // per-wave lists: for each of the kTlQW queries of a pass, distances [capS], original indices [capS], output [outS]
  int *wbase = tl_lds + L.wave0 + wave * L.per_wave;
  const int cap = kTlCapMul * K, cap3 = 3 * K;
  const int lane_ry = lane % 3, lane_rz = (lane / 3) % 3;  // run r = lane < 9: row (y0 + r % 3, z0 + r / 3)

  const int nshare = q_hi - q_lo;
  for (int chunk0 = 0; chunk0 < nshare; chunk0 += kTlQChunk) {
    const int cn = nshare - chunk0 < kTlQChunk ? nshare - chunk0 : kTlQChunk;
    if (chunk0 > 0) {
      __syncthreads();  // the previous round's records have been consumed
#pragma unroll
      for (int u = 0; u < kTlPT; ++u) {
        const int r = qpos[u] - chunk0;
        if (qpos[u] >= 0 && r >= 0 && r < kTlQChunk) {
          const int i = u * kTlThreads + tid;
          const float x = q[i * 3 + 0], y = q[i * 3 + 1], z = q[i * 3 + 2];
          int cxy, cz;
          const int c = cell_of(x, y, z, cxy, cz);
          qrec[r] = make_float4(x, y, z, __int_as_float(i));
          qcell[r] = make_int2(c | (cz << 16), cxy);
        }
      }
    }
    if (tid == 0) s_ticket = 0;
    __syncthreads();  // the scatter is complete, this round's records are staged
    // tasks: runs of <= kTlQT queries of one cell, aligned to multiples of kTlQT inside the cell
    bool start = false;
    int tk = 0;
    if (tid < cn) {
      const int c = qcell[tid].x & 0xffff;
      const int g = q_lo + chunk0 + tid;              // position in the cloud's cell-ordered query sequence
      const int st = c == c_lo ? q_lo : q_end[c - 1];
      const int o = (g - st) % kTlQT;
      start = o == 0 || tid == 0;
      int nn = kTlQT - o;
      const int left = q_end[c] - g;                  // queries of this cell from here on
      nn = nn < left ? nn : left;
      nn = nn < cn - tid ? nn : cn - tid;
      tk = tid | (nn << 16);
    }
    const unsigned long long sm_ = __ballot(start);
    if (lane == 0) s_wave[0][wave] = (int)__popcll(sm_);
    __syncthreads();
    int woff = 0, ntasks = 0;
    for (int ww = 0; ww < kTlWaves; ++ww) {
      const int v = s_wave[0][ww];
      if (ww < wave) woff += v;
      ntasks += v;
    }
    ntasks = __builtin_amdgcn_readfirstlane(ntasks);
    if (start) tasks[woff + prefix_popc(sm_)] = tk;
    __syncthreads();

    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&s_ticket, 1);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t >= ntasks) break;
      const int tkv = __builtin_amdgcn_readfirstlane(tasks[t]);
      const int r0 = tkv & 0xffff, n = tkv >> 16;
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
      // A window of <= 64 * kTlWin records (the rule at the metric shape: ~320) stays in registers for every query of
      // the task; a larger one is streamed again for each pair of queries.  A position past the window's end holds a
      // NaN coordinate: its distances are NaN and never "in radius" -- no per-candidate guard below.
      const bool resident = T <= CL3D_WAVE * kTlWin;
      float4 sp[kTlWin];
      auto load_window = [&](int p0) {
#pragma unroll
        for (int v = 0; v < kTlWin; ++v) {
          const int p = p0 + v * CL3D_WAVE + lane;
          const int pc = p < T ? p : T - 1;
          sp[v] = sorted[TL_WINDOW_POS(pc)];
          if (p >= T) sp[v].x = __builtin_nanf("");
        }
      };
      if (resident && T > 0) load_window(0);

      for (int u0 = 0; u0 < n; u0 += kTlQW) {
        const int nq = n - u0 < kTlQW ? n - u0 : kTlQW;  // uniform
        int jq[kTlQW];
        float qxs[kTlQW], qys[kTlQW], qzs[kTlQW];
        int cnt[kTlQW];
#pragma unroll
        for (int u = 0; u < kTlQW; ++u) {
          const float4 qq = qrec[r0 + u0 + (u < nq ? u : 0)];
          jq[u] = __builtin_amdgcn_readfirstlane(__float_as_int(qq.w));
          // an unused query slot of the pass gets a NaN coordinate: its distances are NaN and never "in radius"
          qxs[u] = u < nq ? tl_uniform(qq.x) : __builtin_nanf("");
          qys[u] = tl_uniform(qq.y);
          qzs[u] = tl_uniform(qq.z);
          cnt[u] = 0;
        }
        const tl_v2f qx2 = {qxs[0], qxs[1]}, qy2 = {qys[0], qys[1]}, qz2 = {qzs[0], qzs[1]};
        // ---- candidates: every in-radius candidate of the window leaves (distance, original index) in the query's LDS
        // list (unordered: the selection below restates the reference's order-dependent rule), one ballot + prefix
        // count per query and 64 candidates
        for (int p0 = 0; p0 < T; p0 += CL3D_WAVE * kTlWin) {
          if (!resident) load_window(p0);
#pragma unroll
          for (int v = 0; v < kTlWin; ++v) {
            if (p0 + v * CL3D_WAVE >= T) break;  // uniform
            const tl_v2f d2 = tl_dist2_pair(qx2, qy2, qz2, sp[v].x, sp[v].y, sp[v].z);
#pragma unroll
            for (int u = 0; u < kTlQW; ++u) {
              const float d = u == 0 ? d2.x : d2.y;
              const bool hit = d < radius2;
              const unsigned long long m = __ballot(hit);
              const int c0 = cnt[u];
              const int c1 = c0 + (int)__popcll(m);  // wave-uniform
              if (c1 <= cap && hit) {                // an overflowing list is abandoned
                int *dst = wbase + u * (2 * L.capS + L.outS) + c0 + prefix_popc(m);
                dst[0] = __float_as_int(d);
                dst[L.capS] = __float_as_int(sp[v].w);
              }
              cnt[u] = c1;
            }
          }
        }
        tl_wave_sync();

#pragma unroll
        for (int u = 0; u < kTlQW; ++u) {
          if (u >= nq) continue;
          const int j = jq[u];
          const float qxu = qxs[u], qyu = qys[u], qzu = qzs[u];
          const int S = __builtin_amdgcn_readfirstlane(cnt[u]);
          int *oi = idx + ((size_t)b * M + j) * K;
          int *om = idx_mask + ((size_t)b * M + j) * K;
          int *lbase = wbase + u * (2 * L.capS + L.outS);
          float *ld = reinterpret_cast<float *>(lbase);
          int *li = lbase + L.capS;
          int *so = lbase + 2 * L.capS;
          int c = S;
          if (S > cap3) {
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
                  const float d2 = dist2(qxu, qyu, qzu, rec.x, rec.y, rec.z);
                  const bool hit = p < T && d2 < radius2 && __float_as_int(rec.w) < mid;
                  below += (int)__popcll(__ballot(hit));
                }
                if (below >= cap3) hi = mid;
                else lo = mid;
              }
              int fill = 0;
              for (int p0 = 0; p0 < T; p0 += CL3D_WAVE) {
                const int p = p0 + lane;
                const float4 rec = sorted[TL_WINDOW_POS(p < T ? p : T - 1)];
                const float d2 = dist2(qxu, qyu, qzu, rec.x, rec.y, rec.z);
                const bool hit = p < T && d2 < radius2;
                const int orig = __float_as_int(rec.w);
                if (hit) {
                  const unsigned long long ke = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)orig;
                  key = ke < key ? ke : key;
                }
                const bool take = hit && orig < hi;
                const unsigned long long m = __ballot(take);
                if (take) {
                  const int e = fill + prefix_popc(m);
                  ld[e] = d2;
                  li[e] = orig;
                }
                fill += (int)__popcll(m);
              }
              nlist = cap3;  // == fill: original indices are distinct
              tl_wave_sync();
            } else {
              for (int e = lane; e < S; e += CL3D_WAVE) {
                const unsigned long long ke = ((unsigned long long)__float_as_uint(ld[e]) << 32) | (unsigned)li[e];
                key = ke < key ? ke : key;
              }
            }
            key = wave_min_u64(key);
            const int gidx = (int)(unsigned)(key & 0xffffffffull);
            // the 3K smallest original indices, moved to the front of the list in index order: every lane takes its
            // entries into registers and ranks them by original index against the whole list, then -- all reads done --
            // the entries with rank < 3K are written back at their rank
            float my_d[kTlIdxRounds];
            int my_i[kTlIdxRounds], my_r[kTlIdxRounds];
#pragma unroll
            for (int t2 = 0; t2 < kTlIdxRounds; ++t2) {
              const int e = t2 * CL3D_WAVE + lane;
              const bool on = e < nlist;
              my_d[t2] = on ? ld[e] : 0.f;
              my_i[t2] = on ? li[e] : 0x7fffffff;
              my_r[t2] = 0;
            }
            for (int f = 0; f < nlist; ++f) {
              const int kf = li[f];
#pragma unroll
              for (int t2 = 0; t2 < kTlIdxRounds; ++t2) my_r[t2] += kf < my_i[t2] ? 1 : 0;
            }
            tl_wave_sync();
#pragma unroll
            for (int t2 = 0; t2 < kTlIdxRounds; ++t2) {
              if (t2 * CL3D_WAVE + lane < nlist && my_r[t2] < cap3) {
                ld[my_r[t2]] = my_d[t2];
                li[my_r[t2]] = my_i[t2];
              }
            }
            tl_wave_sync();
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
          // four at a time from LDS (the list ends in four all-ones sentinels: no tail loop).  Equal distances give
          // equal ranks and leave a hole in ranks [0, min(c, K+1)); a hole is detected below and the exact ranking
          // redoes the (rare) list.
          unsigned *lb = reinterpret_cast<unsigned *>(ld);
          const int need = c < K + 1 ? c : K + 1;
          if (lane < 4) lb[c + lane] = 0xffffffffu;
          for (int i = lane; i < need; i += CL3D_WAVE) so[i] = -1;
          tl_wave_sync();
          for (int e0 = 0; e0 < c; e0 += CL3D_WAVE) {
            const int e = e0 + lane;
            const bool on = e < c;
            const unsigned my = on ? lb[e] : 0u;
            int rank = 0;
            for (int f = 0; f < c; f += 4) {
              const uint4 k4 = *reinterpret_cast<const uint4 *>(lb + f);
              rank += k4.x < my ? 1 : 0;
              rank += k4.y < my ? 1 : 0;
              rank += k4.z < my ? 1 : 0;
              rank += k4.w < my ? 1 : 0;
            }
            if (on && rank <= K) so[rank] = li[e];
          }
          tl_wave_sync();
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
            tl_wave_sync();
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
        tl_wave_sync();
      }
    }
  }
}

bool ball_query_tile_applicable(int M, int N, int K) {
  if (N < 512 || M < 64 || N > kTlMaxPts || M > kTlMaxPts || K < 1) return false;
  if (kTlCapMul * K > kTlIdxRounds * CL3D_WAVE) return false;
  return (size_t)tl_layout(N, K).total * sizeof(int) <= 158 * 1024;
}

int ball_query_tile(const float *query_xyz, const float *support_xyz, const int *query_mask,
                    const int *support_mask, int B, int M, int N, float radius, int K, int *idx, int *idx_mask,
                    hipStream_t st) {
  if (B > 65535) return fail(CL3D_E_UNSUPPORTED, "ball_query: B exceeds grid.y limit");
  const size_t lds = (size_t)tl_layout(N, K).total * sizeof(int);
  static std::atomic<unsigned long long> granted{0};
  int rc = lds_opt_in(granted, reinterpret_cast<const void *>(bq_tile_kernel), 158 * 1024, "ball_query");
  if (rc != CL3D_OK) return rc;
  // workgroups per cloud: one per CU over the whole batch, each with at least 64 queries
  int parts = 256 / (B > 0 ? B : 1);
  const int most = M / 64;
  parts = parts > most ? most : parts;
  parts = parts < 1 ? 1 : (parts > 64 ? 64 : parts);
  hipLaunchKernelGGL(bq_tile_kernel, dim3(parts, B), dim3(kTlThreads), lds, st, query_xyz, support_xyz, query_mask,
                     support_mask, M, N, radius, radius * radius, K, idx, idx_mask);
  return check_launch("cl3d_masked_ordered_ball_query(tile)");
}

}  // namespace cl3d

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
// Tunables: the defaults are what ships; scripts/micro/bq_variants.py builds this file with other values
// (-DCL3D_BQ_MICRO adds a C entry point) and times them against each other on the GPU box.
#ifndef CL3D_TL_QT
#define CL3D_TL_QT 4
#endif
#ifndef CL3D_TL_PHASE
#define CL3D_TL_PHASE 0   // timing experiments only: 1 = stop after the prep, 2 = after the candidate pass, 3 = before the stores
#endif
#ifndef CL3D_TL_RANK
#define CL3D_TL_RANK 1    // 1 = both queries of a pass ranked in lock step; 0 = one after the other (ties by write collisions either way)
#endif
#ifndef CL3D_TL_MINWAVES
#define CL3D_TL_MINWAVES 4  // waves per SIMD the register budget is set for (5 -> 96 VGPRs: room for a second kernel's waves)
#endif
constexpr int kTlQT = CL3D_TL_QT;                 // queries per task (of one cell): they share one window read
constexpr int kTlQW = 2;                          // queries tested per candidate pass (one packed-FP32 chain)
constexpr int kTlCapMul = 6;                      // candidate list holds kTlCapMul*K entries per query
#ifndef CL3D_TL_FLAT
#define CL3D_TL_FLAT 6
#endif
constexpr int kTlFlat = CL3D_TL_FLAT;             // window batches (of 64) whose positions a wave keeps in registers
constexpr int kTlIdxRounds = 4;                   // rank-by-index rounds held in registers: kTlCapMul*K <= 256
constexpr int kTlRankRounds = 2;                  // rank-by-distance rounds held in registers: 3*K <= 128

// both queries of a pass: CL3D_TL_PK = 1 a packed pair (v_pk_*_f32, rounds 4-6; scripts/micro/kernel_variants.py "bq_packed"),
// 0 (shipped since round 6, session 66) two scalar chains -- the packed chain never had fewer cycles (a packed FP32
// instruction takes two slots on gfx950; round 4 said so of its own gain), the scalar one measures the same or 1 % faster
// (67.0-67.3 against 67.5-68.2 us alone, the replayed step 0.2807 against 0.2817 ms, alternating), returns the same bits (it
// IS cl3d::dist2), and leaves no packed operand in the kernel for the fault of DESIGN 6 to find -- the candidate's y comes
// out of ds_read_b128 in the high half of a register pair, exactly the operand that read zeros in the gather pass.
#ifndef CL3D_TL_PK
#define CL3D_TL_PK 0
#endif
#if CL3D_TL_PK
typedef float tl_v2f __attribute__((ext_vector_type(2)));
#else
struct tl_v2f { float x, y; };
#endif

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
#if !CL3D_TL_PK
  tl_v2f r;  // (the scalar chain itself: cl3d::dist2)
  r.x = dist2(qx.x, qy.x, qz.x, x, y, z);
  r.y = dist2(qx.y, qy.y, qz.y, x, y, z);
  return r;
#else
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
#endif
}

template <bool SAME>  // SAME: the queries ARE the support points (every non-strided layer): one set of registers for both roles
__global__ __launch_bounds__(kTlThreads, CL3D_TL_MINWAVES) void bq_tile_kernel(const float *__restrict__ query_xyz,
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
#pragma unroll
  for (int u = 0; u < kTlPT; ++u) {
    if constexpr (SAME) {
      qx_[u] = px[u];
      qy_[u] = py[u];
      qz_[u] = pz[u];
    } else {
      const int i = u * kTlThreads + tid;
      const int ic = i < M ? i : M - 1;
      qx_[u] = q[ic * 3 + 0];
      qy_[u] = q[ic * 3 + 1];
      qz_[u] = q[ic * 3 + 2];
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
    if constexpr (SAME) {
      cs[u] = cq[u] = cell_of(px[u], py[u], pz[u], cqxy[u], cqz[u]);
    } else {
      int dxy, dz;
      cs[u] = cell_of(px[u], py[u], pz[u], dxy, dz);
      cq[u] = cell_of(qx_[u], qy_[u], qz_[u], cqxy[u], cqz[u]);
    }
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
#if CL3D_TL_PHASE == 1
  return;
#endif

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
      // the cell's candidate window: 3x3 (y,z) rows of <= 3 x-adjacent cells, each one contiguous run of `sorted`.
      // Lane r < 9 holds run r's start and length; the passes below walk the runs one after the other, 64 records at a
      // time (position = start + lane: no search for the run a window position falls into -- round 3 spent a third of
      // the candidate pass's vector instructions and most of its branches on that -- at the price of a run's last
      // batch being partly empty).
      int run_a = 0, run_n = 0;
      {
        const int2 cc = qcell[r0];
        const int cz = __builtin_amdgcn_readfirstlane(cc.x >> 16);
        const int cx = __builtin_amdgcn_readfirstlane(cc.y & 0xffff), cy = __builtin_amdgcn_readfirstlane(cc.y >> 16);
        const int x0 = cx > 0 ? cx - 1 : 0, x1 = cx + 1 < nx ? cx + 1 : nx - 1;
        const int y0 = cy > 0 ? cy - 1 : 0, y1 = cy + 1 < ny ? cy + 1 : ny - 1;
        const int z0 = cz > 0 ? cz - 1 : 0, z1 = cz + 1 < nz ? cz + 1 : nz - 1;
        const int yy = y0 + lane_ry, zz = z0 + lane_rz;
        if (lane < 9 && yy <= y1 && zz <= z1) {
          const int row = nx * (yy + ny * zz);
          const int ca = row + x0, cb = row + x1 + 1;  // cells [ca, cb): starts are the previous cell's end
          run_a = ca > 0 ? s_end[ca - 1] : 0;
          run_n = s_end[cb - 1] - run_a;
        }
      }
      // The window laid end to end is T records; when they fit kTlFlat batches (the rule: ~320 at the metric shape) every
      // lane works out, once per task, where its kTlFlat window positions lie in `sorted` -- which run a position falls
      // into is a count of run starts at or below it (eight compares against broadcast prefixes, no branches), the
      // run's offset a cross-lane read -- and the candidate passes of the task's queries read whole batches of 64
      // records from those positions.  Larger windows are walked run by run (for_window).
      int win_pos[kTlFlat];
      int T;
      {
        int inc = run_n;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          const int v = __shfl_up(inc, o, 64);
          if (lane >= o) inc += v;
        }
        T = __builtin_amdgcn_readlane(inc, 8);
        const int pe = inc - run_n;   // lane r < 9: exclusive prefix of run r in the window's index space
        const int dl = run_a - pe;    //             start(run r) - prefix
        int rr[kTlFlat];
#pragma unroll
        for (int v = 0; v < kTlFlat; ++v) rr[v] = 0;
        if (T <= CL3D_WAVE * kTlFlat) {
#pragma unroll
          for (int r = 1; r < 9; ++r) {
            const int per = __builtin_amdgcn_readlane(pe, r);
#pragma unroll
            for (int v = 0; v < kTlFlat; ++v) rr[v] += (v * CL3D_WAVE + lane >= per) ? 1 : 0;
          }
        }
#pragma unroll
        for (int v = 0; v < kTlFlat; ++v) {
          const int pw = v * CL3D_WAVE + lane;
          win_pos[v] = (pw < T ? pw : (T > 0 ? T - 1 : 0)) + __shfl(dl, rr[v], CL3D_WAVE);
        }
      }
      const bool flat = T <= CL3D_WAVE * kTlFlat;
      // f(record, live) for every record of the window, 64 at a time; a lane past a run's end sees live == false.
      // Three runs' first batches are requested together (one LDS round trip for ~100 records); what a run holds
      // beyond 64 records (dense clouds) follows one batch at a time.
      auto for_window = [&](auto &&f) {
#pragma unroll 1
        for (int g = 0; g < 9; g += 3) {
          int ra[3], rn[3];
          float4 rec[3];
#pragma unroll
          for (int t3 = 0; t3 < 3; ++t3) {
            ra[t3] = __builtin_amdgcn_readlane(run_a, g + t3);
            rn[t3] = __builtin_amdgcn_readlane(run_n, g + t3);
          }
#pragma unroll
          for (int t3 = 0; t3 < 3; ++t3)  // (an absent run has start 0, length 0: record 0 is read and ignored)
            rec[t3] = sorted[ra[t3] + (lane < rn[t3] ? lane : (rn[t3] > 0 ? rn[t3] - 1 : 0))];
#pragma unroll
          for (int t3 = 0; t3 < 3; ++t3)
            if (rn[t3] > 0) f(rec[t3], lane < rn[t3]);
#pragma unroll 1
          for (int t3 = 0; t3 < 3; ++t3) {
            const int ra2 = __builtin_amdgcn_readlane(run_a, g + t3), rn2 = __builtin_amdgcn_readlane(run_n, g + t3);
            for (int o = CL3D_WAVE; o < rn2; o += CL3D_WAVE) {
              const int pp = o + lane;
              f(sorted[ra2 + (pp < rn2 ? pp : rn2 - 1)], pp < rn2);
            }
          }
        }
      };

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
        auto test = [&](float4 rec, bool live) {
          // (a lane past the end gets a NaN coordinate: its distances are NaN and never "in radius")
          const tl_v2f d2 = tl_dist2_pair(qx2, qy2, qz2, live ? rec.x : __builtin_nanf(""), rec.y, rec.z);
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
              dst[L.capS] = __float_as_int(rec.w);
            }
            cnt[u] = c1;
          }
        };
        if (flat) {
#pragma unroll
          for (int v0 = 0; v0 < kTlFlat; v0 += 3) {  // three batches in flight per LDS round trip
            if (v0 * CL3D_WAVE >= T) break;           // uniform
            float4 rec[3];
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) rec[t3] = sorted[win_pos[v0 + t3]];
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3)
              if ((v0 + t3) * CL3D_WAVE < T) test(rec[t3], (v0 + t3) * CL3D_WAVE + lane < T);
          }
        } else {
          for_window(test);
        }
        tl_wave_sync();

#if CL3D_TL_PHASE == 2
        continue;
#endif
        // ---- selection (rare): more than 3K in-radius candidates -> the 3K smallest original indices, the strict
        // minimum patched into the last slot; returns the length of the list the ranking works on
        auto select = [&](int u) -> int {
          const float qxu = qxs[u], qyu = qys[u], qzu = qzs[u];
          const int S = __builtin_amdgcn_readfirstlane(cnt[u]);
          int *lbase = wbase + u * (2 * L.capS + L.outS);
          float *ld = reinterpret_cast<float *>(lbase);
          int *li = lbase + L.capS;
          if (S <= cap3) return S;
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
              for_window([&](float4 rec, bool live) {
                const float d2 = dist2(qxu, qyu, qzu, rec.x, rec.y, rec.z);
                const bool hit = live && d2 < radius2 && __float_as_int(rec.w) < mid;
                below += (int)__popcll(__ballot(hit));
              });
              if (below >= cap3) hi = mid;
              else lo = mid;
            }
            int fill = 0;
            for_window([&](float4 rec, bool live) {
              const float d2 = dist2(qxu, qyu, qzu, rec.x, rec.y, rec.z);
              const bool hit = live && d2 < radius2;
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
            });
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
          return cap3;
        };
        // exact ranking by (distance, original index) of a list whose distances tie: ranks < K go to `so`
        auto rank_exact = [&](int u, int c) {
          int *lbase = wbase + u * (2 * L.capS + L.outS);
          const float *ld = reinterpret_cast<const float *>(lbase);
          const int *li = lbase + L.capS;
          int *so = lbase + 2 * L.capS;
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
        };
        // first K of the ranked list to the caller's arrays; wrap-around padding for a short list
        auto store = [&](int u, int c, int first) {  // first = so[lane] (lane < K), read by the caller
          const int j = jq[u];
          int *oi = idx + ((size_t)b * M + j) * K;
          int *om = idx_mask + ((size_t)b * M + j) * K;
          const int *so = wbase + u * (2 * L.capS + L.outS) + 2 * L.capS;
          const int qmk = qm[j];
#if CL3D_TL_PHASE == 3
          if (qmk == 0x7fffffff) oi[lane] = first + c;  // (timing experiment: everything but the stores)
          return;
#endif
          if (c >= K) {  // uniform, the common case: a full list, no wrap-around padding (and no integer modulo)
            if (lane < K) {
              oi[lane] = first;
              om[lane] = qmk != 0 ? 1 : 0;
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
        };
        // ---- rank by (distance, original index) == stable sort by distance of the index-ordered list.
        // Fast path: rank by the distance alone (d2 >= 0, so its bit pattern orders like the value), keys broadcast
        // four at a time from LDS (the list ends in four all-ones sentinels: no tail loop).
#if CL3D_TL_RANK == 1
        // Both queries of the pass in lock step (one LDS round trip serves two independent chains).  A tie in distance
        // gives two entries the same rank: both write the same slot of `so`, one of them reads back another entry's
        // index -- the collision sends the (rare) list to the exact ranking.  Only ranks < K matter: a tie further out
        // cannot change the first K.
        int cq2[kTlQW];
#pragma unroll
        for (int u = 0; u < kTlQW; ++u) cq2[u] = u < nq ? select(u) : 0;
        unsigned *lb0 = reinterpret_cast<unsigned *>(wbase);
        unsigned *lb1 = reinterpret_cast<unsigned *>(wbase + (2 * L.capS + L.outS));
        if (lane < 4) {
          lb0[cq2[0] + lane] = 0xffffffffu;
          lb1[cq2[1] + lane] = 0xffffffffu;
        }
        tl_wave_sync();
        int rk[kTlQW][kTlRankRounds], mi[kTlQW][kTlRankRounds];
        {
          // round 0 of both lists: one merged loop over the common length, then the longer list's tail
          const int c0 = cq2[0], c1 = cq2[1];
          const unsigned my0 = lane < c0 ? lb0[lane] : 0u, my1 = lane < c1 ? lb1[lane] : 0u;
          mi[0][0] = lane < c0 ? reinterpret_cast<const int *>(lb0)[L.capS + lane] : -1;
          mi[1][0] = lane < c1 ? reinterpret_cast<const int *>(lb1)[L.capS + lane] : -1;
          int r0 = 0, r1 = 0;
          const int cm = c0 < c1 ? c0 : c1;
          int f = 0;
          for (; f < cm; f += 4) {
            const uint4 a4 = *reinterpret_cast<const uint4 *>(lb0 + f);
            const uint4 b4 = *reinterpret_cast<const uint4 *>(lb1 + f);
            r0 += a4.x < my0 ? 1 : 0; r0 += a4.y < my0 ? 1 : 0; r0 += a4.z < my0 ? 1 : 0; r0 += a4.w < my0 ? 1 : 0;
            r1 += b4.x < my1 ? 1 : 0; r1 += b4.y < my1 ? 1 : 0; r1 += b4.z < my1 ? 1 : 0; r1 += b4.w < my1 ? 1 : 0;
          }
          for (int g = f; g < c0; g += 4) {
            const uint4 a4 = *reinterpret_cast<const uint4 *>(lb0 + g);
            r0 += a4.x < my0 ? 1 : 0; r0 += a4.y < my0 ? 1 : 0; r0 += a4.z < my0 ? 1 : 0; r0 += a4.w < my0 ? 1 : 0;
          }
          for (int g = f; g < c1; g += 4) {
            const uint4 b4 = *reinterpret_cast<const uint4 *>(lb1 + g);
            r1 += b4.x < my1 ? 1 : 0; r1 += b4.y < my1 ? 1 : 0; r1 += b4.z < my1 ? 1 : 0; r1 += b4.w < my1 ? 1 : 0;
          }
          rk[0][0] = lane < c0 ? r0 : K;
          rk[1][0] = lane < c1 ? r1 : K;
        }
#pragma unroll
        for (int u = 0; u < kTlQW; ++u) {  // entries 64 .. 3K - 1 of a long list (uniform, uncommon)
          rk[u][1] = K;
          mi[u][1] = -1;
          const int c = cq2[u];
          if (c > CL3D_WAVE) {
            const unsigned *lb = u == 0 ? lb0 : lb1;
            const int e = CL3D_WAVE + lane;
            const bool on = e < c;
            const unsigned my = on ? lb[e] : 0u;
            mi[u][1] = on ? reinterpret_cast<const int *>(lb)[L.capS + e] : -1;
            int r = 0;
            for (int g = 0; g < c; g += 4) {
              const uint4 k4 = *reinterpret_cast<const uint4 *>(lb + g);
              r += k4.x < my ? 1 : 0; r += k4.y < my ? 1 : 0; r += k4.z < my ? 1 : 0; r += k4.w < my ? 1 : 0;
            }
            rk[u][1] = on ? r : K;
          }
        }
        int *so0 = wbase + 2 * L.capS, *so1 = wbase + (2 * L.capS + L.outS) + 2 * L.capS;
#pragma unroll
        for (int t2 = 0; t2 < kTlRankRounds; ++t2) {
          if (rk[0][t2] < K) so0[rk[0][t2]] = mi[0][t2];
          if (rk[1][t2] < K) so1[rk[1][t2]] = mi[1][t2];
        }
        tl_wave_sync();
        bool bad0 = false, bad1 = false;
#pragma unroll
        for (int t2 = 0; t2 < kTlRankRounds; ++t2) {
          if (rk[0][t2] < K) bad0 = bad0 || so0[rk[0][t2]] != mi[0][t2];
          if (rk[1][t2] < K) bad1 = bad1 || so1[rk[1][t2]] != mi[1][t2];
        }
        int first0 = lane < K ? so0[lane] : 0, first1 = lane < K ? so1[lane] : 0;
        if (__ballot(bad0) != 0ull) {  // uniform: ties in distance -> exact (distance, original index) ranking
          rank_exact(0, cq2[0]);
          first0 = lane < K ? so0[lane] : 0;
        }
        if (__ballot(bad1) != 0ull) {
          rank_exact(1, cq2[1]);
          first1 = lane < K ? so1[lane] : 0;
        }
        store(0, cq2[0], first0);
        if (nq > 1) store(1, cq2[1], first1);
#else
        // One query at a time, same collision rule.
#pragma unroll
        for (int u = 0; u < kTlQW; ++u) {
          if (u >= nq) continue;
          const int c = select(u);
          int *lbase = wbase + u * (2 * L.capS + L.outS);
          const int *li = lbase + L.capS;
          int *so = lbase + 2 * L.capS;
          unsigned *lb = reinterpret_cast<unsigned *>(lbase);
          if (lane < 4) lb[c + lane] = 0xffffffffu;
          tl_wave_sync();
          int rk[kTlRankRounds], mi[kTlRankRounds];
#pragma unroll
          for (int t2 = 0; t2 < kTlRankRounds; ++t2) {
            rk[t2] = K;
            mi[t2] = -1;
            if (t2 * CL3D_WAVE < c) {  // uniform; the second round only for a list of more than 64
              const int e = t2 * CL3D_WAVE + lane;
              const bool on = e < c;
              const unsigned my = on ? lb[e] : 0u;
              mi[t2] = on ? li[e] : -1;
              int rank = 0;
#pragma unroll 2
              for (int f = 0; f < c; f += 4) {
                const uint4 k4 = *reinterpret_cast<const uint4 *>(lb + f);
                rank += k4.x < my ? 1 : 0;
                rank += k4.y < my ? 1 : 0;
                rank += k4.z < my ? 1 : 0;
                rank += k4.w < my ? 1 : 0;
              }
              rk[t2] = on ? rank : K;
            }
          }
#pragma unroll
          for (int t2 = 0; t2 < kTlRankRounds; ++t2)
            if (rk[t2] < K) so[rk[t2]] = mi[t2];
          tl_wave_sync();
          bool bad = false;
#pragma unroll
          for (int t2 = 0; t2 < kTlRankRounds; ++t2)
            if (rk[t2] < K) bad = bad || so[rk[t2]] != mi[t2];
          int first = lane < K ? so[lane] : 0;
          if (__ballot(bad) != 0ull) {  // uniform: ties in distance -> exact (distance, original index) ranking
            rank_exact(u, c);
            first = lane < K ? so[lane] : 0;
          }
          store(u, c, first);
        }
#endif
        tl_wave_sync();
      }
    }
  }
}

bool ball_query_tile_applicable(int M, int N, int K) {
  if (N < 512 || M < 64 || N > kTlMaxPts || M > kTlMaxPts || K < 1) return false;
  if (kTlCapMul * K > kTlIdxRounds * CL3D_WAVE || 3 * K > kTlRankRounds * CL3D_WAVE || K > CL3D_WAVE) return false;
  return (size_t)tl_layout(N, K).total * sizeof(int) <= 158 * 1024;
}

int ball_query_tile(const float *query_xyz, const float *support_xyz, const int *query_mask,
                    const int *support_mask, int B, int M, int N, float radius, int K, int *idx, int *idx_mask,
                    hipStream_t st) {
  if (B > 65535) return fail(CL3D_E_UNSUPPORTED, "ball_query: B exceeds grid.y limit");
  const size_t lds = (size_t)tl_layout(N, K).total * sizeof(int);
  static std::atomic<unsigned long long> granted{0}, granted2{0};
  const bool same = query_xyz == support_xyz && M == N;
  int rc = lds_opt_in(granted, reinterpret_cast<const void *>(bq_tile_kernel<true>), 158 * 1024, "ball_query");
  if (rc == CL3D_OK) rc = lds_opt_in(granted2, reinterpret_cast<const void *>(bq_tile_kernel<false>), 158 * 1024, "ball_query");
  if (rc != CL3D_OK) return rc;
  // workgroups per cloud: one per CU over the whole batch, each with at least 64 queries
  int parts = 256 / (B > 0 ? B : 1);
  const int most = M / 64;
  parts = parts > most ? most : parts;
  parts = parts < 1 ? 1 : (parts > 64 ? 64 : parts);
  if (same)
    hipLaunchKernelGGL(bq_tile_kernel<true>, dim3(parts, B), dim3(kTlThreads), lds, st, query_xyz, support_xyz, query_mask,
                       support_mask, M, N, radius, radius * radius, K, idx, idx_mask);
  else
    hipLaunchKernelGGL(bq_tile_kernel<false>, dim3(parts, B), dim3(kTlThreads), lds, st, query_xyz, support_xyz, query_mask,
                       support_mask, M, N, radius, radius * radius, K, idx, idx_mask);
  return check_launch("cl3d_masked_ordered_ball_query(tile)");
}

}  // namespace cl3d

#ifdef CL3D_BQ_MICRO
// scripts/micro/bq_variants.py: this file alone as a shared library, one variant of the tunables per build
extern "C" int cl3d_bq_micro(const float *q, const float *s, const int *qm, const int *sm, int B, int M, int N, float radius,
                             int K, int *idx, int *idx_mask, void *stream) {
  if (!cl3d::ball_query_tile_applicable(M, N, K)) return -1;
  return cl3d::ball_query_tile(q, s, qm, sm, B, M, N, radius, K, idx, idx_mask, (hipStream_t)stream);
}
#endif

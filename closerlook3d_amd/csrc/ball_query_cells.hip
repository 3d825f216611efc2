// ball_query_cells.hip -- masked ordered ball query through a uniform cell grid (gfx950).
//
// Same results, bit for bit, as ball_query.hip's exhaustive scan (and therefore as the reference,
// masked_ordered_ball_query_gpu.cu:11-96), but a query only looks at the support points of the 27
// cells around it instead of all N:
//
//   prep   (one workgroup per cloud)  bounding box of the valid support points, cell size h >= radius
//          (grown until the grid has <= kMaxCells cells), counting sort of the support points by cell
//          into `sorted` (float4 {x,y,z, original index}), counting sort of the query indices by cell,
//          and a task table: every task = up to QW queries of ONE cell.
//   query  (persistent waves)  a wave takes a task; its QW queries share the same 9 contiguous runs
//          of `sorted` (3 cells in x are adjacent in the cell order), so lanes stream candidates as
//          coalesced float4 and every candidate is tested against all QW queries, exactly like the
//          exhaustive kernel -- with ~N/13 candidates instead of N at the metric shape.
//
// The reference's semantics depend on the ORIGINAL index order ("first 3*nsample in-radius points by
// support index", strict running minimum patched into the last slot, stable sort by distance), and the
// cell order is not the index order.  So the wave collects ALL in-radius candidates S (distance,
// original index) in LDS and then restates the rule order-independently:
//   |S| <= 3K : the candidate list is S;
//   |S| >  3K : the 3K smallest original indices of S (rank by index), and if the (first-occurrence)
//               minimum of S is not among them it replaces the one with the largest index;
//   result    : the list ranked by (distance, original index) -- what a stable sort of the
//               index-ordered list by distance gives -- first K, wrap-around padding.
// |S| > kCap*K candidates do not fit the LDS list: the query is flagged and redone by the exhaustive
// kernel (ball_query.hip, flag-filtered launch), so the result is exact for any density.
//
// A support point with float d2 < r^2 can never fall outside the 27 cells: h = r*(1+2e-4) leaves four
// orders of magnitude more slack than the rounding of the cell coordinates (grid <= kMaxCells cells).
#include "ball_query.h"

namespace cl3d {

constexpr int kMaxCells = 8192;
constexpr int kBqQW = 2;   // queries per task / wave (measured at the metric shape: 2 -> 102 us, 4 -> 122 us, 1 -> 137 us)
constexpr int kCapMul = 6; // LDS candidate list holds kCapMul*K entries per query
constexpr int kBqBatch = 3;  // candidate float4 loads in flight per lane
constexpr int kPrepU = 4;    // point sweeps in flight per thread in the prep kernel

struct BqGrid {
  float ox, oy, oz, inv_h;
  int nx, ny, nz, ncells;
  int ntasks, nv, pad0, pad1;
};

// One task = up to kBqQW queries of one cell, together with that cell's candidate window: the <= 9
// contiguous runs of `sorted` (3x3 (y,z) rows, <= 3 x-adjacent cells each) laid end to end.  run_pe[r] is
// the exclusive prefix of run r in that index space, run_delta[r] = start(r) - run_pe[r].  Everything a
// wave needs to start streaming candidates arrives with one 96-byte uniform load.
struct BqTask {
  int q0, n, total, cell;
  int run_pe[9], run_delta[9];
  int pad[2];
};

// workspace layout (per call), see cl3d_workspace_bytes(CL3D_OP_BALL_QUERY)
struct BqWorkspace {
  BqGrid *grid;        // [B]
  float4 *sorted;      // [B,N]
  float4 *qsorted;     // [B,M] queries grouped by cell: {x,y,z, original query index}
  BqTask *tasks;       // [B, M/QW + kMaxCells + 1]
  int *overflow;       // [B,M]
  int max_tasks;
};

__host__ __device__ inline size_t bq_align(size_t x) { return (x + 255) & ~(size_t)255; }

inline size_t bq_workspace_bytes(int B, int N, int M) {
  const size_t max_tasks = (size_t)(M + kBqQW - 1) / kBqQW + kMaxCells + 1;
  return bq_align(sizeof(BqGrid) * B) + bq_align(sizeof(float4) * (size_t)B * N) +
         bq_align(sizeof(float4) * (size_t)B * M) +
         bq_align(sizeof(BqTask) * (size_t)B * max_tasks) + bq_align(sizeof(int) * (size_t)B * M);
}

inline BqWorkspace bq_carve(void *ws, int B, int N, int M) {
  BqWorkspace w;
  char *p = static_cast<char *>(ws);
  w.max_tasks = (M + kBqQW - 1) / kBqQW + kMaxCells + 1;
  w.grid = reinterpret_cast<BqGrid *>(p); p += bq_align(sizeof(BqGrid) * B);
  w.sorted = reinterpret_cast<float4 *>(p); p += bq_align(sizeof(float4) * (size_t)B * N);
  w.qsorted = reinterpret_cast<float4 *>(p); p += bq_align(sizeof(float4) * (size_t)B * M);
  w.tasks = reinterpret_cast<BqTask *>(p); p += bq_align(sizeof(BqTask) * (size_t)B * w.max_tasks);
  w.overflow = reinterpret_cast<int *>(p);
  return w;
}

__device__ __forceinline__ int cell_coord(float x, float o, float inv_h) { return (int)floorf((x - o) * inv_h); }

// prep: gridDim.x 1024-thread workgroups per cloud.  The work is tiny (tens of KB) and entirely latency:
// every __syncthreads-separated phase costs a global round trip, so the phases are merged as far as the
// data dependences allow -- (1) one sweep over the support cloud gives the number of leading valid points
// and the bounding box; (2) support AND query histograms over the cells, side by side in LDS; (3) both
// exclusive scans share their barriers (the third column scanned is "tasks per cell"); (4) the task table
// (with each cell's candidate runs) and both scatters.
// Phases (1)-(3) are cheap and every workgroup of a cloud repeats them (same inputs, same arithmetic, so
// all of them hold the same grid and the same cell starts in LDS -- no inter-block communication); the
// expensive phase (4) is split: workgroup r owns the contiguous cell range that holds the r-th share of
// the points (resp. queries) and writes only those cells' records and tasks.
__global__ __launch_bounds__(1024) void bq_prep_kernel(const float *__restrict__ query_xyz,
                                                       const float *__restrict__ support_xyz,
                                                       const int *__restrict__ support_mask, int M, int N,
                                                       float radius, BqWorkspace w) {
  extern __shared__ int lds_cells[];  // [2][kMaxCells]: support counts/starts/cursors, query counts/starts/cursors
  __shared__ float s_red[6][16];
  __shared__ int s_wave[3][16];
  __shared__ int s_nv;
  __shared__ int s_share[2][2];  // [support|query][first cell, end cell) owned by this workgroup
  int *s_sup = lds_cells, *s_qry = lds_cells + kMaxCells;
  const int b = blockIdx.y;
  const int part = blockIdx.x, nparts = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *s = support_xyz + (size_t)b * N * 3;
  const float *q = query_xyz + (size_t)b * M * 3;
  const int *sm = support_mask + (size_t)b * N;

  // ---- (1) first zero of the mask + bounding box of the unmasked points (a superset of the valid prefix
  // is fine for a search grid; masked-out coordinates never enter it)
  if (tid == 0) s_nv = N;
  for (int c = tid; c < 2 * kMaxCells; c += 1024) lds_cells[c] = 0;
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  int first0 = N;
  // loads are issued kPrepU sweeps at a time and unconditionally (clamped index), so a sweep costs one
  // memory round trip instead of one per dependent load
  for (int base = 0; base < N; base += 1024 * kPrepU) {
    int mk[kPrepU];
    float px[kPrepU], py[kPrepU], pz[kPrepU];
#pragma unroll
    for (int u = 0; u < kPrepU; ++u) {
      const int i = base + u * 1024 + tid;
      const int ic = i < N ? i : N - 1;
      mk[u] = sm[ic];
      px[u] = s[ic * 3 + 0];
      py[u] = s[ic * 3 + 1];
      pz[u] = s[ic * 3 + 2];
    }
#pragma unroll
    for (int u = 0; u < kPrepU; ++u) {
      const int i = base + u * 1024 + tid;
      if (i >= N) continue;
      if (mk[u] == 0) {
        first0 = i < first0 ? i : first0;
        continue;
      }
      mn[0] = px[u] < mn[0] ? px[u] : mn[0]; mx[0] = px[u] > mx[0] ? px[u] : mx[0];
      mn[1] = py[u] < mn[1] ? py[u] : mn[1]; mx[1] = py[u] > mx[1] ? py[u] : mx[1];
      mn[2] = pz[u] < mn[2] ? pz[u] : mn[2]; mx[2] = pz[u] > mx[2] ? pz[u] : mx[2];
    }
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
  __syncthreads();  // s_nv / lds_cells initialised, s_red written
  if (first0 < N) atomicMin(&s_nv, first0);
  __syncthreads();
  const int nv = s_nv;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    mn[a] = s_red[a][0];
    mx[a] = s_red[3 + a][0];
    for (int ww = 1; ww < 16; ++ww) {
      mn[a] = s_red[a][ww] < mn[a] ? s_red[a][ww] : mn[a];
      mx[a] = s_red[3 + a][ww] > mx[a] ? s_red[3 + a][ww] : mx[a];
    }
  }
  // cell size: >= radius with slack, grown until the grid fits kMaxCells (every thread computes the same)
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
      ok = nx > 0 && ny > 0 && nz > 0 && (long long)nx * ny * nz <= kMaxCells;
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
  auto cell_of = [&](float x, float y, float z) {
    int cx = cell_coord(x, mn[0], inv_h), cy = cell_coord(y, mn[1], inv_h), cz = cell_coord(z, mn[2], inv_h);
    cx = cx < 0 ? 0 : (cx >= nx ? nx - 1 : cx);
    cy = cy < 0 ? 0 : (cy >= ny ? ny - 1 : cy);
    cz = cz < 0 ? 0 : (cz >= nz ? nz - 1 : cz);
    return cx + nx * (cy + ny * cz);
  };

  // ---- (2) both histograms
  auto sweep = [&](const float *pts, int n, auto &&fn) {  // batched point loads, see (1)
    for (int base = 0; base < n; base += 1024 * kPrepU) {
      float px[kPrepU], py[kPrepU], pz[kPrepU];
#pragma unroll
      for (int u = 0; u < kPrepU; ++u) {
        const int i = base + u * 1024 + tid;
        const int ic = i < n ? i : n - 1;
        px[u] = pts[ic * 3 + 0];
        py[u] = pts[ic * 3 + 1];
        pz[u] = pts[ic * 3 + 2];
      }
#pragma unroll
      for (int u = 0; u < kPrepU; ++u) {
        const int i = base + u * 1024 + tid;
        if (i < n) fn(i, px[u], py[u], pz[u]);
      }
    }
  };
  // queries == support points (every non-strided layer): one sweep feeds both histograms, and both scatters below
  const bool same = query_xyz == support_xyz && M == N;
  if (same) {
    sweep(s, N, [&](int i, float x, float y, float z) {
      const int cell = cell_of(x, y, z);
      if (i < nv) atomicAdd(&s_sup[cell], 1);
      atomicAdd(&s_qry[cell], 1);
    });
  } else {
    sweep(s, nv, [&](int, float x, float y, float z) { atomicAdd(&s_sup[cell_of(x, y, z)], 1); });
    sweep(q, M, [&](int, float x, float y, float z) { atomicAdd(&s_qry[cell_of(x, y, z)], 1); });
  }
  __syncthreads();

  // ---- (3) three exclusive scans over the cells with shared barriers: support counts, query counts,
  // tasks per cell.  Thread t owns cells [t*per, (t+1)*per).
  const int per = (ncells + 1023) / 1024;  // <= 8
  const int t0 = tid * per;
  int nq_mine[8];
  int sum[3] = {0, 0, 0};
  for (int i = 0; i < 8; ++i) {
    nq_mine[i] = 0;
    if (i < per && t0 + i < ncells) {
      nq_mine[i] = s_qry[t0 + i];
      sum[0] += s_sup[t0 + i];
      sum[1] += nq_mine[i];
      sum[2] += (nq_mine[i] + kBqQW - 1) / kBqQW;
    }
  }
  int incl[3] = {sum[0], sum[1], sum[2]};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl[k], o, 64);
      if (lane >= o) incl[k] += v;
    }
    if (lane == 63) s_wave[k][wave] = incl[k];
  }
  __syncthreads();
  int run[3], total[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int woff = 0, tot = 0;
    for (int ww = 0; ww < 16; ++ww) {
      if (ww < wave) woff += s_wave[k][ww];
      tot += s_wave[k][ww];
    }
    run[k] = woff + incl[k] - sum[k];
    total[k] = tot;
  }
  for (int i = 0; i < 8; ++i) {
    if (i < per && t0 + i < ncells) {
      const int cs_ = s_sup[t0 + i];
      s_sup[t0 + i] = run[0];
      run[0] += cs_;
      s_qry[t0 + i] = run[1];
      run[1] += nq_mine[i];
    }
  }
  __syncthreads();

  // ---- (4) task table and scatters of this workgroup's share of the cells
  if (tid < 4) {  // first cell whose start reaches the share boundary (starts are non-decreasing)
    const int col = tid >> 1, end = tid & 1;
    const int *starts = col == 0 ? s_sup : s_qry;
    const long long tot = col == 0 ? nv : M;
    const int pr = part + end;
    int lo = 0, hi = ncells;
    if (pr <= 0) hi = 0;
    else if (pr >= nparts) lo = ncells;
    else {
      const int target = (int)(tot * pr / nparts);
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (starts[mid] >= target) hi = mid;
        else lo = mid + 1;
      }
    }
    s_share[col][end] = pr <= 0 ? 0 : lo;
  }
  __syncthreads();
  const int sup_lo = s_share[0][0], sup_hi = s_share[0][1], qry_lo = s_share[1][0], qry_hi = s_share[1][1];
  auto start_of = [&](int c) { return c < ncells ? s_sup[c] : nv; };
  BqTask *tasks = w.tasks + (size_t)b * w.max_tasks;
  {
    int t = run[2];
    for (int i = 0; i < 8; ++i) {
      if (!(i < per && t0 + i < ncells) || nq_mine[i] == 0) continue;
      if (t0 + i < qry_lo || t0 + i >= qry_hi) {
        t += (nq_mine[i] + kBqQW - 1) / kBqQW;
        continue;
      }
      // candidate window of this cell (queries that were clamped into the grid get a superset of what
      // their true position needs: outside the grid only the boundary cells can be within reach)
      BqTask tk;
      const int cell = t0 + i;
      const int cx = cell % nx, cy = (cell / nx) % ny, cz = cell / (nx * ny);
      const int x0 = cx > 0 ? cx - 1 : 0, x1 = cx + 1 < nx ? cx + 1 : nx - 1;
      const int y0 = cy > 0 ? cy - 1 : 0, y1 = cy + 1 < ny ? cy + 1 : ny - 1;
      const int z0 = cz > 0 ? cz - 1 : 0, z1 = cz + 1 < nz ? cz + 1 : nz - 1;
      int acc = 0;
#pragma unroll  // keeps the task record in registers (a dynamic index would put it in scratch memory)
      for (int r = 0; r < 9; ++r) {
        const int yy = y0 + r % 3, zz = z0 + r / 3;
        int ra = 0, len = 0;
        if (yy <= y1 && zz <= z1) {
          const int row = nx * (yy + ny * zz);
          ra = start_of(row + x0);
          len = start_of(row + x1 + 1) - ra;
        }
        tk.run_pe[r] = acc;
        tk.run_delta[r] = ra - acc;
        acc += len;
      }
      tk.total = acc;
      tk.cell = cell;
      tk.pad[0] = tk.pad[1] = 0;
      const int qs = s_qry[cell];
      for (int k = 0; k < nq_mine[i]; k += kBqQW) {
        tk.q0 = qs + k;
        tk.n = nq_mine[i] - k < kBqQW ? nq_mine[i] - k : kBqQW;
        tasks[t++] = tk;
      }
    }
  }
  __syncthreads();  // every reader of the start values is done: the arrays become scatter cursors
  float4 *sorted = w.sorted + (size_t)b * N;
  float4 *qsorted = w.qsorted + (size_t)b * M;
  auto scatter = [&](const float *pts, int n, int *cursor, float4 *dst, int c_lo, int c_hi) {  // order inside a cell is irrelevant
    for (int base = 0; base < n; base += 1024 * kPrepU) {
      float px[kPrepU], py[kPrepU], pz[kPrepU];
      int pos[kPrepU];
#pragma unroll
      for (int u = 0; u < kPrepU; ++u) {
        const int i = base + u * 1024 + tid;
        const int ic = i < n ? i : n - 1;
        px[u] = pts[ic * 3 + 0];
        py[u] = pts[ic * 3 + 1];
        pz[u] = pts[ic * 3 + 2];
      }
#pragma unroll
      for (int u = 0; u < kPrepU; ++u)  // all cursor atomics of the batch first, then all stores
      {
        const int cell = cell_of(px[u], py[u], pz[u]);
        pos[u] = (base + u * 1024 + tid < n && cell >= c_lo && cell < c_hi) ? atomicAdd(&cursor[cell], 1) : -1;
      }
#pragma unroll
      for (int u = 0; u < kPrepU; ++u)
        if (pos[u] >= 0) dst[pos[u]] = make_float4(px[u], py[u], pz[u], __int_as_float(base + u * 1024 + tid));
    }
  };
  if (same) {  // one load per point, two cursors
    for (int base = 0; base < N; base += 1024 * kPrepU) {
      float px[kPrepU], py[kPrepU], pz[kPrepU];
      int ps[kPrepU], pq[kPrepU];
#pragma unroll
      for (int u = 0; u < kPrepU; ++u) {
        const int i = base + u * 1024 + tid;
        const int ic = i < N ? i : N - 1;
        px[u] = s[ic * 3 + 0];
        py[u] = s[ic * 3 + 1];
        pz[u] = s[ic * 3 + 2];
      }
#pragma unroll
      for (int u = 0; u < kPrepU; ++u) {
        const int i = base + u * 1024 + tid;
        const int cell = cell_of(px[u], py[u], pz[u]);
        ps[u] = (i < nv && cell >= sup_lo && cell < sup_hi) ? atomicAdd(&s_sup[cell], 1) : -1;
        pq[u] = (i < N && cell >= qry_lo && cell < qry_hi) ? atomicAdd(&s_qry[cell], 1) : -1;
      }
#pragma unroll
      for (int u = 0; u < kPrepU; ++u) {
        const float4 rec = make_float4(px[u], py[u], pz[u], __int_as_float(base + u * 1024 + tid));
        if (ps[u] >= 0) sorted[ps[u]] = rec;
        if (pq[u] >= 0) qsorted[pq[u]] = rec;
      }
    }
  } else {
    scatter(s, nv, s_sup, sorted, sup_lo, sup_hi);
    scatter(q, M, s_qry, qsorted, qry_lo, qry_hi);
  }
  if (tid == 0 && part == 0) {
    BqGrid g;
    g.ox = mn[0]; g.oy = mn[1]; g.oz = mn[2]; g.inv_h = inv_h;
    g.nx = nx; g.ny = ny; g.nz = nz; g.ncells = ncells;
    g.ntasks = total[2]; g.nv = nv; g.pad0 = g.pad1 = 0;
    w.grid[b] = g;
  }
}

// LDS strides of the per-query lists, rounded so every list starts 16-byte aligned (ds_read_b128 in the ranking)
__host__ __device__ inline int bq_pad4(int x) { return (x + 3) & ~3; }
__host__ __device__ inline int bq_lds_ints_per_wave(int K) {
  return kBqQW * (2 * bq_pad4(kCapMul * K) + 2 * bq_pad4(3 * K) + bq_pad4(K + 1));
}

// rank of every list element by (distance, original index), exact for any list: 64-bit keys from LDS
__device__ __forceinline__ void bq_rank_exact(const float *ld, const int *li, int c, int K, int *so, int lane) {
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
}

__global__ __launch_bounds__(256) void bq_query_kernel(const float *__restrict__ query_xyz,
                                                       const int *__restrict__ query_mask, int M, int N,
                                                       float radius2, int K, BqWorkspace w,
                                                       int *__restrict__ idx, int *__restrict__ idx_mask) {
  extern __shared__ int smem[];
  const int cap3 = 3 * K;
  const int cap = kCapMul * K;
  const int capS = bq_pad4(cap), cap3S = bq_pad4(cap3), outS = bq_pad4(K + 1);
  const int b = blockIdx.y;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // LDS carve per wave: cand_d[QW][capS], cand_i[QW][capS], sel_d[QW][cap3S], sel_i[QW][cap3S], out_i[QW][outS]
  int *base = smem + (size_t)wave * bq_lds_ints_per_wave(K);
  float *cand_d = reinterpret_cast<float *>(base);
  int *cand_i = base + kBqQW * capS;
  float *sel_d = reinterpret_cast<float *>(base + 2 * kBqQW * capS);
  int *sel_i = base + 2 * kBqQW * capS + kBqQW * cap3S;
  int *out_i = base + 2 * kBqQW * capS + 2 * kBqQW * cap3S;

  const int ntasks = w.grid[b].ntasks;
  const int *qm = query_mask + (size_t)b * M;
  const float4 *sorted = w.sorted + (size_t)b * N;
  const float4 *qsorted = w.qsorted + (size_t)b * M;
  const BqTask *tasks = w.tasks + (size_t)b * w.max_tasks;
  int *oflow = w.overflow + (size_t)b * M;

  // (a per-cloud ticket counter for dynamic balance was tried: ~2300 returning atomics on one address per cloud
  // serialise at ~60 ns each -- 148 us for the kernel against 53 us with this static walk)
  for (int t = blockIdx.x * 4 + wave; t < ntasks; t += gridDim.x * 4) {
    const BqTask tk = tasks[t];
    const int n = tk.n;
    int jq[kBqQW];
    float qx[kBqQW], qy[kBqQW], qz[kBqQW];
    int cnt[kBqQW];
#pragma unroll
    for (int u = 0; u < kBqQW; ++u) {
      const float4 qq = qsorted[tk.q0 + (u < n ? u : 0)];
      jq[u] = __float_as_int(qq.w);
      // an unused query slot of the task gets a NaN coordinate: its distances are NaN and never "in radius"
      qx[u] = u < n ? qq.x : __builtin_nanf("");
      qy[u] = qq.y;
      qz[u] = qq.z;
      cnt[u] = 0;
    }
    const int T = tk.total;
    int run_pe[9], run_delta[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      run_pe[r] = tk.run_pe[r];
      run_delta[r] = tk.run_delta[r];
    }
    // ---- candidates: every in-radius (distance, original index) pair of the window goes to the LDS list.
    // Lists are unordered (the ranking below restates the reference's order-dependent rule), so a batch of
    // 64 candidates costs one ballot + prefix count per query and nothing else.
    for (int p0 = 0; p0 < T; p0 += CL3D_WAVE * kBqBatch) {
      float4 sp[kBqBatch];
#pragma unroll
      for (int v = 0; v < kBqBatch; ++v) {
        if (p0 + v * CL3D_WAVE >= T) break;  // uniform
        int p = p0 + v * CL3D_WAVE + lane;
        p = p < T ? p : T - 1;
        int delta = run_delta[0];
#pragma unroll
        for (int r = 1; r < 9; ++r) delta = p >= run_pe[r] ? run_delta[r] : delta;
        sp[v] = sorted[p + delta];
      }
#pragma unroll
      for (int v = 0; v < kBqBatch; ++v) {
        if (p0 + v * CL3D_WAVE >= T) break;  // uniform
        const bool live = p0 + v * CL3D_WAVE + lane < T;
        const int orig = __float_as_int(sp[v].w);
#pragma unroll
        for (int u = 0; u < kBqQW; ++u) {
          const float d2 = dist2(qx[u], qy[u], qz[u], sp[v].x, sp[v].y, sp[v].z);
          const bool hit = live && (d2 < radius2);
          const unsigned long long m = __ballot(hit);
          const int c0 = cnt[u];
          const int c1 = c0 + (int)__popcll(m);  // wave-uniform
          if (c1 <= cap && hit) {  // a list that would overflow is abandoned: that query is redone exhaustively
            const int pos = c0 + prefix_popc(m);
            cand_d[u * capS + pos] = d2;
            cand_i[u * capS + pos] = orig;
          }
          cnt[u] = c1;
        }
      }
    }
    // LDS traffic below is wave-private and in program order; the fence keeps the compiler from reordering
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();

#pragma unroll
    for (int u = 0; u < kBqQW; ++u) {
      if (u >= n) continue;
      const int j = jq[u];
      const int S = __builtin_amdgcn_readfirstlane(cnt[u]);
      int *oi = idx + ((size_t)b * M + j) * K;
      int *om = idx_mask + ((size_t)b * M + j) * K;
      if (S > cap) {  // too dense for the LDS list: the exhaustive kernel redoes this query
        if (lane == 0) oflow[j] = 1;
        continue;
      }
      if (lane == 0) oflow[j] = 0;
      float *ld = cand_d + u * capS;
      int *li = cand_i + u * capS;
      int c = S;
      if (S > cap3) {
        // first-occurrence strict minimum == smallest (d2, original index) of S
        unsigned long long key = ~0ull;
        for (int e = lane; e < S; e += CL3D_WAVE) {
          const unsigned long long ke = ((unsigned long long)__float_as_uint(ld[e]) << 32) | (unsigned)li[e];
          key = ke < key ? ke : key;
        }
        key = wave_min_u64(key);
        const int gidx = (int)(unsigned)(key & 0xffffffffull);
        // the 3K smallest original indices, written in index order
        float *sd = sel_d + u * cap3S;
        int *si = sel_i + u * cap3S;
        for (int e = lane; e < S; e += CL3D_WAVE) {
          const int ie = li[e];
          int r = 0;
#pragma unroll 8
          for (int f = 0; f < S; ++f) r += (li[f] < ie) ? 1 : 0;
          if (r < cap3) {
            sd[r] = ld[e];
            si[r] = ie;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (gidx > si[cap3 - 1]) {  // uniform: the minimum was cut off -> it takes the last slot
          if (lane == 0) {
            si[cap3 - 1] = gidx;
            sd[cap3 - 1] = __uint_as_float((unsigned)(key >> 32));
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        ld = sd;
        li = si;
        c = cap3;
      }
      // ---- rank by (distance, original index) == stable sort by distance of the index-ordered list.
      // Fast path: rank by the distance alone (d2 >= 0, so its bit pattern orders like the value): keys are
      // broadcast four at a time from LDS, 2 VALU instructions per comparison.  Equal distances give equal
      // ranks and leave a hole in ranks [0, min(c, K+1)); a hole is detected below and the exact 64-bit
      // ranking redoes the (rare) list.
      int *so = out_i + u * outS;
      const int need = c < K + 1 ? c : K + 1;
      for (int i = lane; i < need; i += CL3D_WAVE) so[i] = -1;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
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
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      bool hole = false;
      for (int i = lane; i < need; i += CL3D_WAVE) hole = hole || so[i] < 0;
      if (__ballot(hole) != 0ull) {  // uniform
        bq_rank_exact(ld, li, c, K, so, lane);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      const int qmk = qm[j];
      if (c >= K) {  // uniform, the common case: a full list, no wrap-around padding (and no integer modulo)
        for (int i = lane; i < K; i += CL3D_WAVE) {
          oi[i] = so[i];
          om[i] = qmk != 0 ? 1 : 0;
        }
      } else {
        for (int i = lane; i < K; i += CL3D_WAVE) {
          int v = 0, mk = 0;
          if (c > 0) {
            v = so[i < c ? i : i % c];
            mk = (i < c && qmk != 0) ? 1 : 0;
          }
          oi[i] = v;
          om[i] = mk;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace cl3d

namespace cl3d {

size_t ball_query_cells_workspace(int B, int N, int M) { return bq_workspace_bytes(B, N, M); }

bool ball_query_cells_applicable(int M, int N, int K) {
  const size_t lds = (size_t)4 * bq_lds_ints_per_wave(K) * sizeof(int);
  return N >= 512 && M >= 64 && lds <= 64 * 1024;
}

int ball_query_cells(const float *query_xyz, const float *support_xyz, const int *query_mask,
                     const int *support_mask, int B, int M, int N, float radius, int K, int *idx,
                     int *idx_mask, void *ws, size_t ws_bytes, hipStream_t st) {
  if (ws == nullptr || ws_bytes < bq_workspace_bytes(B, N, M))
    return fail(CL3D_E_WORKSPACE, "ball_query: workspace %zu < %zu", ws_bytes, bq_workspace_bytes(B, N, M));
  if (B > 65535) return fail(CL3D_E_UNSUPPORTED, "ball_query: B exceeds grid.y limit");
  BqWorkspace w = bq_carve(ws, B, N, M);
  // 64 KiB of cell arrays + the static reduction scratch: above the 64 KiB a kernel gets by default
  static std::atomic<unsigned long long> prep_granted{0};
  int rc_lds = lds_opt_in(prep_granted, reinterpret_cast<const void *>(bq_prep_kernel), 2 * kMaxCells * sizeof(int), "ball_query");
  if (rc_lds != CL3D_OK) return rc_lds;
  // workgroups per cloud in the prep kernel: enough to split the scatter / task-table work without flooding the chip
  int parts = 256 / (B > 0 ? B : 1);
  parts = parts < 1 ? 1 : (parts > 8 ? 8 : parts);
  hipLaunchKernelGGL(bq_prep_kernel, dim3(parts, B), dim3(1024), 2 * kMaxCells * sizeof(int), st, query_xyz, support_xyz,
                     support_mask, M, N, radius, w);
  const size_t lds = (size_t)4 * bq_lds_ints_per_wave(K) * sizeof(int);
  // one task per wave: M/QW tasks if every cell held a multiple of QW queries, plus one per cell with a remainder
  // (the host does not know the cell count: M/8 extra covers the metric shape's ~12 %; waves without a task leave at
  // once, waves beyond the cap walk the table with the grid's stride)
  int gx = ceil_div(ceil_div(M, kBqQW) + ceil_div(M, 8), 4);
  gx = gx > 1024 ? 1024 : gx;
  // (measured per-cloud grids at the metric shape, B = 16: 128 -> 71.9 us, 256 -> 64.8, 384 -> 63.2, 512 -> 63.0, 640 -> 60.8)
  hipLaunchKernelGGL(bq_query_kernel, dim3(gx, B), dim3(256), lds, st, query_xyz, query_mask, M, N, radius * radius, K, w, idx, idx_mask);
  int rc = check_launch("cl3d_masked_ordered_ball_query(cells)");
  if (rc != CL3D_OK) return rc;
  // queries too dense for the LDS list were flagged; the exhaustive kernel redoes exactly those
  return ball_query_exhaustive(query_xyz, support_xyz, query_mask, support_mask, B, M, N, radius, K, idx, idx_mask,
                               w.overflow, st);
}

}  // namespace cl3d

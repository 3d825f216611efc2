// fused_pwmlp.hip -- PointWiseMLP ('dp_fi_df', one conv+BN+ReLU layer, max reduction) for gfx950.
//
// Reference (models/local_aggregation_operators.py:288-301): gather [B,C,M,K], build
// x = cat[rel(3), f_centre(C), f_nbr - f_centre(C)] (68.7 MB per cloud at the metric shape), 1x1
// Conv2d (131 -> C_out) over all B*M*K positions (2.2 GFLOP per cloud), BatchNorm2d, ReLU, max over K.
//
// The contraction is linear, so it factors through the points instead of the (point, neighbour) pairs:
//     W x = W_r rel + (W_c - W_d) f_centre + W_d f_nbr
// With G = W_d F and H = (W_c - W_d) F computed ONCE PER POINT (a plain [B*N, C] x [C, 2*C_out] GEMM,
// K times fewer flops than the reference's contraction; done by the caller with a library GEMM so
// autograd also provides dF and dW from dG, dH), every pre-activation is
//     y[b,o,j,k] = W_r[o,:] . rel[b,j,k] + H[b, idx[b,j,0], o] + G[b, idx[b,j,k], o]
// i.e. one point-major row gather + 3 FMAs.  `ght` is [B, N, 2*C_out]: row i = [G_i | H_i].
//
// Nothing of size B*C*M*K is ever stored, and in training the rows are gathered only TWICE (once
// query-major, once support-major), because everything else is algebra on per-(query, channel) values:
//   * BN+ReLU is monotone in y (z = scale*y + shift, sign(scale) = sign(gamma) is known before the batch
//     statistics are), so max_k ReLU(z) = ReLU(z(y*)) with y* = max_k y (min_k y for gamma < 0): the pass
//     that accumulates the batch statistics also finds y* and its slot k*;
//   * the BatchNorm backward is affine in y (dy = A dz [k = k*] + Bc + D y), so sum_k dy and
//     sum_k dy * rel follow from sum_k y, dz and the per-channel sums S_a = sum y*rel_a, R_a = sum rel_a.
// Passes:
//   TRAIN     (gather, query-major) per channel: sum y, sum y^2, S_a, R_a (double partials);
//             per (query, channel): y*, k*, sum_k y;
//   APPLY     (element-wise) out = ReLU(scale*y* + shift), transposed to channel-major through LDS
//   FWD       (gather; inference with running statistics) the same output in one pass
//   BWD_ROWS  (element-wise) dz at the arg-max (ReLU gate), d beta = sum dz, d gamma = sum dz*xhat,
//             T_a = sum dz*rel_a(k*) (double partials); then d W_r = A T + Bc R + D S per channel;
//             dz and t* leave channel-major for the next pass
//   HIT       (stream + LDS scatter) hit[i,c] = sum of dz over the (query, channel) pairs with t* = i
//   BWD_SUPPORT (gather, support-major through the LDS-staged CSR inverse of idx)
//             dG_i = D (W_r . sum rel + sum H[centre] + |S_i| G_i) + |S_i| Bc + A hit_i,
//             dH_i = sum over queries centred on i of (D sum_k y + K Bc + A dz).  Ordered, no float atomics
//             in global memory.
//   plus the weight plumbing around the per-point library GEMM (split_weight / merge_weight_grad).
#include "fused_common.h"
#include <stdlib.h>
#include <type_traits>

// The TRAIN walk's arithmetic: 0 (shipped since round 6, session 65) = scalar FMAs, two channels' chains interleaved by the
// compiler; 1 = packed pairs (v_pk_fma_f32, rounds 4-6: scripts/micro/kernel_variants.py "train_packed").  A packed FP32
// instruction occupies the vector pipe for two slots on gfx950, so the packed walk never had fewer cycles, only fewer
// instructions -- and its four-deep dependent chains stalled the issue (round 5's ISA account: 34 % of the pass).  Scalar,
// same box, alternating runs: TRAIN 66.2-66.5 -> 62.6-63.4 us, the replayed step 0.2842-0.2856 -> 0.2809-0.2824 ms, 110
// VGPRs against 115, the same bits (each packed half WAS the scalar operation); and no packed operand is left in the
// pass for the fault of DESIGN 6 to find (profiles/r06/session65_summary.txt).
#ifndef CL3D_TRAIN_PK
#define CL3D_TRAIN_PK 0
#endif

namespace cl3d {

enum { PW_TRAIN = 0, PW_FWD = 1 };
constexpr int kPartialW = 8;   // doubles per (block, channel) in the partial-sum buffers

struct PwArgs {
  const float *query_xyz, *support_xyz;
  const int *idx;
  const float *ght;  // [B,N,2Co]
  const float *wr;   // [Co,3]
  const float *v0, *v1, *v2;  // per-channel vectors: TRAIN gamma | FWD scale,shift | SUPPORT A,Bc,D
  float *out_t;                    // FWD
  int out_channel_major;
  float *ystar_t, *sy_t;           // TRAIN: [B,M,Co]
  unsigned char *kstar_out;        // TRAIN / FWD
  const float *hit_cm;             // SUPPORT: [B,Co,N] sum of arg-max dz per support point (channel-major)
  const float *dz_t;               // SUPPORT: [B,M,Co] gated upstream gradient (point-major copy left by bwd_rows)
  const float4 *qtab;              // SUPPORT: [B,M] {query coordinates, centre index idx[j, 0]} left by bwd_rows
  const float *sy_in;              // SUPPORT: [B,M,Co]
  double *partial;                 // [gridDim.x, Co, kPartialW]
  const int *inv_off, *inv_slots;
  float *dght;                     // SUPPORT: [B,N,2Co]
  int B, N, M, K, Co;
  int L, QW, chunks;
  float inv_radius;
  unsigned kmagic;  // ceil(2^32 / K), see div_k
};

// y = W_r . rel + H[centre] + G[neighbour], as one chain of fused multiply-adds started on the centre term (one
// instruction per term; the TRAIN pass evaluates the same chain on pre-signed operands, see there)
__device__ __forceinline__ float pw_preact(const float w[3], float rx, float ry, float rz, float hc, float g) {
  float t = __builtin_fmaf(w[0], rx, hc);
  t = __builtin_fmaf(w[1], ry, t);
  t = __builtin_fmaf(w[2], rz, t);
  return t + g;
}

// (div_k / div_magic: cl3d_common.h)
typedef float pw_f2 __attribute__((ext_vector_type(2)));

// A scalar that every lane pair of a packed instruction reads, in a register pair of its own with the value in the pair's
// LOW dword (the other dword is never read: both result lanes take the low one, `op_sel_hi` = 0 on that operand).  Left
// to itself the compiler feeds v_pk_fma_f32 the slot record's registers as ds_read_b128 delivered them, and for rel.z --
// the record's last dword, the HIGH half of an aligned pair -- sets `op_sel` on the operand (the low result lane reads
// the pair's high dword).  That form returned 0.0 to the low lane, in the wave's last sixteen lanes, a few hundred
// times per launch whenever a dense bf16 MFMA contraction -- the engine's or the vendor library's, in this process or
// in another one -- ran on the same CUs: sum y / y* / the batch statistics wrong by W_r[c][z] * rel.z of one slot in the
// even channels (round 6, sessions 37-62; DESIGN 6 "What may run beside a bf16 contraction").  Without `op_sel`: 0 wrong
// elements in 240 launches against 60-100 % of the launches with it.  tests/test_isa_packed_operands.py holds the
// library to it.
__device__ __forceinline__ pw_f2 pk_low(float x) {
  pw_f2 p = __builtin_shufflevector((pw_f2)(x), (pw_f2)(x), 0, -1);  // (element 1 left undefined: one move, not two)
  asm("" : "+v"(p));
  return __builtin_shufflevector(p, p, 0, 0);
}

// query-major gather passes.  Persistent blocks: tile = 4*QW*QPG queries of one cloud (every lane group walks QPG
// queries of a tile one after the other).
// V == 4 is only used when Co % 4 == 0 and V == 1 rows are single elements, so a lane with c0 < Co always
// owns a full vector: every row access is one global_load_dwordx4 / dword.
//
// What the round-3 rewrite changed (the pass was 71-76 us at the metric shape, half of it staging latency):
//   * a slot record is {byte offset of the neighbour's row in the cloud's ght, rel}: the gather address is one
//     32-bit add on a uniform base (was a 64-bit multiply-add per slot and lane);
//   * staging is batched: a thread issues the index loads of all its slots, then all coordinate loads -- two
//     dependent round trips per tile whatever the tile size (was two per 256 slots), and QPG = 2 halves the tiles
//     (barriers, round trips) per slot;
//   * the centre's H row is fetched together with the first batch of G rows (it was a round trip of its own in
//     front of every query's walk);
//   * full batches carry no per-slot guard: with the guard the compiler sank one of the four gathers of a batch
//     behind a branch and waited for it alone (two dependent round trips per batch instead of one);
//   * (PIPE) the walk in two rotating pairs of slots, see the walk below: 69.1 -> 64.5 us.
// Tried and dropped in round 3: the tile's slot records copied out of a table built once per geometry on the ball
// query's stream ([B,M,K] x 16 bytes, 19.5 us to build) instead of chasing idx -> coordinates while staging: the pass
// itself 64.5 -> 68.1 us and the replayed step 0.365-0.372 -> 0.394-0.396 ms -- the 33.5 MB table costs more to
// stream than the two dependent round trips it replaces (coordinates and indices are L2-resident, the table is not).
template <int MODE, int V, int KB, int WPE, int QPG, bool PIPE = false>  // KB = row gathers in flight per lane, WPE = waves per SIMD to fit
__global__ __launch_bounds__(256, WPE) void pwmlp_query_kernel(PwArgs a) {
  extern __shared__ float4 lds4[];
  const int K = a.K, Co = a.Co, M = a.M, N = a.N, L = a.L, QW = a.QW;
  const int TG = 4 * QW;       // lane groups of the block
  const int TQ = TG * QPG;     // queries per tile
  const int row = 2 * Co;
  const unsigned rowb = (unsigned)row * 4u;
  // per-query rows are K+1 records long: the lane groups of a wave read the same slot of different queries at
  // once, and rows of K float4 (512 B at K = 32) would put all of them in the same LDS banks
  const int KS = K + 1;
  float4 *slot4 = lds4;  // [TQ][KS] {row byte offset, rx, ry, rz}
  float4 *qtile = slot4 + TQ * KS;  // [TQ] coordinates of the tile's queries
  double *red = reinterpret_cast<double *>(qtile + TQ);  // [4 waves][L*V*NACC]
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int g = lane / L, cl = lane - g * L;
  const bool lane_on = g < QW;
  const int tiles_per_cloud = (M + TQ - 1) / TQ;
  const int ntiles = a.B * tiles_per_cloud;
  constexpr int NACC = MODE == PW_TRAIN ? 8 : 0;
  constexpr int kStage = 4;  // slots staged per thread and round

  // channel chunks are spread over gridDim.y: the deep layers of a backbone have few queries and many channels
  // (16 clouds x 16 queries x 1152 channels), and a grid over query tiles alone leaves the chip empty there
  for (int ch = blockIdx.y; ch < a.chunks; ch += gridDim.y) {
    const int c0 = (ch * L + cl) * V;
    const bool chan_on = lane_on && c0 < Co;
    const unsigned lane_off = (unsigned)c0 * 4u;
    // per-channel constants.  TRAIN works on y' = sgn * y (see below) and keeps only the pre-signed weights;
    // FWD keeps the weights and the folded BatchNorm scale / shift.
    constexpr int NW = MODE == PW_TRAIN ? 1 : V;
    float ws[V][3], sgn[V], c_v0[NW], c_v1[NW];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int c = chan_on ? c0 + v : 0;
      sgn[v] = 1.f;
      if constexpr (MODE == PW_TRAIN) sgn[v] = a.v0[c] < 0.f ? -1.f : 1.f;
      ws[v][0] = sgn[v] * a.wr[c * 3 + 0];
      ws[v][1] = sgn[v] * a.wr[c * 3 + 1];
      ws[v][2] = sgn[v] * a.wr[c * 3 + 2];
      if constexpr (MODE != PW_TRAIN) {
        c_v0[v] = a.v0 ? a.v0[c] : 0.f;
        c_v1[v] = a.v1 ? a.v1[c] : 0.f;
      }
    }
    (void)c_v0;
    (void)c_v1;

    // TRAIN: a lane sums its queries' {sum y, sum y^2, S_0..2, R_0..2} in float (a few hundred terms at most
    // between flushes); every kFlushTiles tiles the lane groups of a wave are folded in a fixed order and
    // group 0 adds the result to the wave's double accumulators in LDS, one slot per (wave, channel) -- no
    // atomics, and half the VGPRs of double register accumulators, which the gather loop's occupancy needs
    constexpr int kFlushTiles = 8 / QPG;  // <= 8 queries x K slots per running float sum
    constexpr int NF = NACC > 0 ? 5 : 1;
    // (CL3D_TRAIN_PK = 1 only -- V == 4: the running sums live as two packed pairs per quantity, the TRAIN walk below is v_pk_fma_f32 / v_pk_add_f32
    // on channel pairs: per-element IEEE, the same bits as the scalar chain, two channels per VALU slot)
    constexpr bool PK = CL3D_TRAIN_PK && V == 4 && MODE == PW_TRAIN;
    constexpr int VA = PK ? 2 : V;
    using acc_t = typename std::conditional<PK, pw_f2, float>::type;
    acc_t accf[NF][VA];
    float accr[3] = {0.f, 0.f, 0.f};
    auto acc_get = [&](int p, int v) -> float {
      if constexpr (PK) return accf[p][v >> 1][v & 1];
      else return accf[p][v];
    };
    auto acc_zero = [&]() {
#pragma unroll
      for (int p = 0; p < NF; ++p)
#pragma unroll
        for (int v = 0; v < VA; ++v) accf[p][v] = acc_t(0.f);
    };
    acc_zero();
    int since_flush = 0;
    const int LVN = L * V * NACC;
    if constexpr (NACC > 0) {
      __syncthreads();
      for (int t = threadIdx.x; t < 4 * LVN; t += 256) red[t] = 0.0;
      __syncthreads();
    }
    // lane group gg of the wave = lanes [gg*L, gg*L+L); L need not be a power of two
    auto fold_groups = [&](float x) {
      float tot = x;
      for (int gg = 1; gg * L + L <= CL3D_WAVE; ++gg) tot += __shfl(x, cl + gg * L, CL3D_WAVE);
      return tot;  // meaningful in group 0
    };
    auto flush = [&]() {  // every lane of the wave takes part in the shuffles
      float f[NF][V], fr[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) fr[p] = fold_groups(accr[p]);
#pragma unroll
      for (int p = 0; p < NF; ++p)
#pragma unroll
        for (int v = 0; v < V; ++v) f[p][v] = fold_groups(acc_get(p, v)) * (p >= 2 ? sgn[v] : 1.f);  // S_a held primed
      if (g == 0) {
        double *mine = red + (size_t)wave * LVN + cl * V * NACC;
#pragma unroll
        for (int v = 0; v < V; ++v) {
#pragma unroll
          for (int p = 0; p < NF; ++p) mine[v * NACC + p] += (double)f[p][v];
#pragma unroll
          for (int p = 0; p < 3; ++p) mine[v * NACC + NF + p] += (double)fr[p];
        }
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) accr[p] = 0.f;
      acc_zero();
      since_flush = 0;
    };
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      int b, tq;
      decode_tile(tile, a.B, tiles_per_cloud, b, tq);
      const int j0 = tq * TQ;
      const float *q = a.query_xyz + (size_t)b * M * 3;
      const float *s = a.support_xyz + (size_t)b * N * 3;
      const int *idxb = a.idx + (size_t)b * M * K;
      __syncthreads();  // previous tile's readers are done with slot4
      float4 qreg = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((int)threadIdx.x < TQ) {  // the tile's query coordinates: requested first, parked in LDS below
        const int j = j0 + (int)threadIdx.x < M ? j0 + (int)threadIdx.x : M - 1;
        qreg = make_float4(q[j * 3 + 0], q[j * 3 + 1], q[j * 3 + 2], 0.f);
      }
      for (int t0 = 0; t0 < TQ * K; t0 += 256 * kStage) {
        // round trip 1: the neighbour indices of this thread's slots; round trip 2: the neighbours' coordinates.
        // Only the indices and 12 coordinates are live across the waits (the gather loop's accumulators stay in
        // registers around this: occupancy is what the pass lives on).  The slot counter goes through an empty asm
        // so that the compiler does not hoist every round's index arithmetic out of the tile loop into registers.
        int iv[kStage];
        float sx[kStage], sy[kStage], sz[kStage];
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
          int t = t0 + u * 256 + (int)threadIdx.x;
          asm volatile("" : "+v"(t));
          const int tc = t < TQ * K ? t : TQ * K - 1;
          const int jq = div_k(tc, a.kmagic, K);
          const int j = j0 + jq < M ? j0 + jq : M - 1;
          iv[u] = idxb[(size_t)j * K + (tc - jq * K)];
        }
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
          sx[u] = s[iv[u] * 3 + 0];
          sy[u] = s[iv[u] * 3 + 1];
          sz[u] = s[iv[u] * 3 + 2];
        }
        if (t0 == 0) {
          if ((int)threadIdx.x < TQ) qtile[threadIdx.x] = qreg;
          __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
          int t = t0 + u * 256 + (int)threadIdx.x;
          asm volatile("" : "+v"(t));
          if (t >= TQ * K) continue;
          const int jq = div_k(t, a.kmagic, K);
          const int k = t - jq * K;
          float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
          if (j0 + jq < M) {
            const float4 qq = qtile[jq];
            r = make_float4(__uint_as_float((unsigned)iv[u] * rowb), (sx[u] - qq.x) * a.inv_radius,
                            (sy[u] - qq.y) * a.inv_radius, (sz[u] - qq.z) * a.inv_radius);
          }
          slot4[jq * KS + k] = r;
        }
      }
      __syncthreads();
      const char *rows = reinterpret_cast<const char *>(a.ght + (size_t)b * N * row);
      auto ldrow = [&](unsigned off) { return load_row<V>(reinterpret_cast<const float *>(rows + off)); };
#pragma unroll 1
      for (int qp = 0; qp < QPG; ++qp) {
      const int jq = qp * TG + wave * QW + g;
      const int j = j0 + jq;
      const bool q_on = chan_on && j < M;
      if (MODE != PW_TRAIN && !q_on) continue;
      const float4 *myslots = slot4 + (q_on ? jq : 0) * KS;
      const size_t orow = ((size_t)b * M + j) * Co + c0;

      // batch 0 of the walk together with the centre's H row (centre = nearest neighbour = slot 0, reference :290)
      float4 sr[KB];
      Vec<V> gr[KB], hc;
#pragma unroll
      for (int v = 0; v < V; ++v) hc.v[v] = 0.f;
#pragma unroll
      for (int u = 0; u < KB; ++u) sr[u] = myslots[u < K ? u : K - 1];
      if (q_on) {
#pragma unroll
        for (int u = 0; u < KB; ++u) gr[u] = ldrow(__float_as_uint(sr[u].x) + lane_off);
        hc = ldrow(__float_as_uint(sr[0].x) + (unsigned)Co * 4u + lane_off);
      }
      // walk: batch 0 is in flight; full batches carry no guards, the tail (K % KB slots) is guarded
      auto walk = [&](auto &&slot) {
        if constexpr (PIPE && KB == 4) {
          // two pairs in rotation: while a pair's two slots are multiplied out, the other pair's rows are in flight
          // (requested right after that pair was consumed) -- the wave itself overlaps gather latency with arithmetic
          // instead of leaving that to the other three waves of its SIMD.  ONE branch-free loop body (K % 4 == 0), so
          // that the waits are exact counts (vmcnt(2)) and not the vmcnt(0) a join of guarded paths forces: the last
          // round re-requests the list's last rows (L1 hits) instead of branching around the refill.
          {  // (launched for K % 4 == 0 only)
#pragma unroll 1
            for (int k0 = 0; k0 < K; k0 += 4) {
              // (scheduling fences: left to itself the compiler hoists the refills and runs out of registers)
              slot(std::false_type{}, k0, sr[0], gr[0]);
              slot(std::false_type{}, k0 + 1, sr[1], gr[1]);
              __builtin_amdgcn_sched_barrier(0);
              const int ka = k0 + 4 < K ? k0 + 4 : K - 2;
              sr[0] = myslots[ka];
              sr[1] = myslots[ka + 1];
              gr[0] = ldrow(__float_as_uint(sr[0].x) + lane_off);
              gr[1] = ldrow(__float_as_uint(sr[1].x) + lane_off);
              __builtin_amdgcn_sched_barrier(0);
              slot(std::false_type{}, k0 + 2, sr[2], gr[2]);
              slot(std::false_type{}, k0 + 3, sr[3], gr[3]);
              __builtin_amdgcn_sched_barrier(0);
              const int kb2 = k0 + 6 < K ? k0 + 6 : K - 2;
              sr[2] = myslots[kb2];
              sr[3] = myslots[kb2 + 1];
              gr[2] = ldrow(__float_as_uint(sr[2].x) + lane_off);
              gr[3] = ldrow(__float_as_uint(sr[3].x) + lane_off);
              __builtin_amdgcn_sched_barrier(0);
            }
            return;
          }
        }
        if (K >= KB) {
#pragma unroll
          for (int u = 0; u < KB; ++u) {
            if (u == 0) slot(std::true_type{}, 0, sr[0], gr[0]);
            else slot(std::false_type{}, u, sr[u], gr[u]);
          }
        } else {
#pragma unroll
          for (int u = 0; u < KB; ++u) {
            if (u == 0) slot(std::true_type{}, 0, sr[0], gr[0]);
            else if (u < K) slot(std::false_type{}, u, sr[u], gr[u]);
          }
        }
        int k0 = KB;
        for (; k0 + KB <= K; k0 += KB) {
#pragma unroll
          for (int u = 0; u < KB; ++u) sr[u] = myslots[k0 + u];
#pragma unroll
          for (int u = 0; u < KB; ++u) gr[u] = ldrow(__float_as_uint(sr[u].x) + lane_off);
#pragma unroll
          for (int u = 0; u < KB; ++u) slot(std::false_type{}, k0 + u, sr[u], gr[u]);
        }
        if (k0 < K) {
#pragma unroll
          for (int u = 0; u < KB; ++u) sr[u] = myslots[k0 + u < K ? k0 + u : K - 1];
#pragma unroll
          for (int u = 0; u < KB; ++u) gr[u] = ldrow(__float_as_uint(sr[u].x) + lane_off);
#pragma unroll
          for (int u = 0; u < KB; ++u)
            if (k0 + u < K) slot(std::false_type{}, k0 + u, sr[u], gr[u]);
        }
      };

      if constexpr (MODE == PW_TRAIN) {
        // everything inside the slot loop works on y' = sgn * y (sgn = sign(gamma) = sign(scale): which extreme of
        // y wins the max).  Negation is exact, so with w' = sgn*w and h' = sgn*h,  y' = fma(g, sgn, w'.rel + h')
        // is sgn * y bit for bit, sum y'^2 = sum y^2, and sum y / sum y*rel are sgn times the primed sums --
        // the sign costs nothing per slot.  sum y^2, S_a and R_a go straight into the lane's running float sums
        // (flushed to double every kFlushTiles tiles); only sum_k y, which is also a per-query output, has its own
        // registers -- the pass lives on its register budget.
        float s1[V], best[V], hs[V];
        int kb[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
          s1[v] = 0.f;
          // (the rotating-pair walk treats slot 0 like every other slot: any finite y beats -inf)
          best[v] = PIPE ? -__builtin_huge_valf() : 0.f;
          kb[v] = 0;
          hs[v] = sgn[v] * hc.v[v];
        }
        if constexpr (PK) {
        if (q_on) {
          pw_f2 w0[2], w1[2], w2[2], sg[2], h2[2], s2[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            w0[h] = (pw_f2){ws[2 * h][0], ws[2 * h + 1][0]};
            w1[h] = (pw_f2){ws[2 * h][1], ws[2 * h + 1][1]};
            w2[h] = (pw_f2){ws[2 * h][2], ws[2 * h + 1][2]};
            sg[h] = (pw_f2){sgn[2 * h], sgn[2 * h + 1]};
            h2[h] = (pw_f2){hs[2 * h], hs[2 * h + 1]};
            s2[h] = (pw_f2)(0.f);
          }
          walk([&](auto first, int k, const float4 &sr_, const Vec<V> &gr_) {
            accr[0] += sr_.y;
            accr[1] += sr_.z;
            accr[2] += sr_.w;
            const pw_f2 rx = pk_low(sr_.y), ry = pk_low(sr_.z), rz = pk_low(sr_.w);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              pw_f2 t = __builtin_elementwise_fma(w0[h], rx, h2[h]);
              t = __builtin_elementwise_fma(w1[h], ry, t);
              t = __builtin_elementwise_fma(w2[h], rz, t);
              const pw_f2 y = __builtin_elementwise_fma((pw_f2){gr_.v[2 * h], gr_.v[2 * h + 1]}, sg[h], t);
              s2[h] += y;
              accf[1][h] = __builtin_elementwise_fma(y, y, accf[1][h]);
              accf[2][h] = __builtin_elementwise_fma(y, rx, accf[2][h]);
              accf[3][h] = __builtin_elementwise_fma(y, ry, accf[3][h]);
              accf[4][h] = __builtin_elementwise_fma(y, rz, accf[4][h]);
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                if (decltype(first)::value || y[e] > best[2 * h + e]) {  // first extreme
                  best[2 * h + e] = y[e];
                  kb[2 * h + e] = k;
                }
              }
            }
          });
          Vec<V> ys, sy;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            s2[h] *= sg[h];
            accf[0][h] += s2[h];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              ys.v[2 * h + e] = sgn[2 * h + e] * best[2 * h + e];
              sy.v[2 * h + e] = s2[h][e];
            }
          }
          store_row<V>(a.ystar_t + orow, ys);
          store_row<V>(a.sy_t + orow, sy);
          *reinterpret_cast<unsigned *>(a.kstar_out + orow) =
              (unsigned)kb[0] | ((unsigned)kb[1] << 8) | ((unsigned)kb[2] << 16) | ((unsigned)kb[3] << 24);
        }
        } else {
        if (q_on) {
        walk([&](auto first, int k, const float4 &sr_, const Vec<V> &gr_) {
          accr[0] += sr_.y;
          accr[1] += sr_.z;
          accr[2] += sr_.w;
#pragma unroll
          for (int v = 0; v < V; ++v) {
            float t = __builtin_fmaf(ws[v][0], sr_.y, hs[v]);
            t = __builtin_fmaf(ws[v][1], sr_.z, t);
            t = __builtin_fmaf(ws[v][2], sr_.w, t);
            const float y = __builtin_fmaf(gr_.v[v], sgn[v], t);  // = sgn * pw_preact(w, rel, hc, g)
            s1[v] += y;
            accf[1][v] = __builtin_fmaf(y, y, accf[1][v]);
            accf[2][v] = __builtin_fmaf(y, sr_.y, accf[2][v]);
            accf[3][v] = __builtin_fmaf(y, sr_.z, accf[3][v]);
            accf[4][v] = __builtin_fmaf(y, sr_.w, accf[4][v]);
            if (decltype(first)::value || y > best[v]) {  // first extreme
              best[v] = y;
              kb[v] = k;
            }
          }
        });
        Vec<V> ys, sy;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          ys.v[v] = sgn[v] * best[v];
          s1[v] *= sgn[v];
          sy.v[v] = s1[v];
          accf[0][v] += s1[v];
        }
        store_row<V>(a.ystar_t + orow, ys);
        store_row<V>(a.sy_t + orow, sy);
        if constexpr (V == 4) {
          *reinterpret_cast<unsigned *>(a.kstar_out + orow) =
              (unsigned)kb[0] | ((unsigned)kb[1] << 8) | ((unsigned)kb[2] << 16) | ((unsigned)kb[3] << 24);
        } else {
          a.kstar_out[orow] = (unsigned char)kb[0];
        }
        }  // q_on
        }  // scalar walk
      } else {  // PW_FWD
        float best[V];
        int kb[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
          best[v] = PIPE ? -__builtin_huge_valf() : 0.f;
          kb[v] = 0;
        }
        walk([&](auto first, int k, const float4 &sr_, const Vec<V> &gr_) {
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const float y = pw_preact(ws[v], sr_.y, sr_.z, sr_.w, hc.v[v], gr_.v[v]);  // sgn = 1 here: ws = w
            float z = __builtin_fmaf(y, c_v0[v], c_v1[v]);
            z = z > 0.f ? z : 0.f;
            if (decltype(first)::value || z > best[v]) {
              best[v] = z;
              kb[v] = k;
            }
          }
        });
        // the operator's output is channel-major [B,Co,M] at the API boundary: written directly (4-byte
        // stores, merged in L2 across the 16 neighbouring queries of the tile) instead of a transpose pass
        _Pragma("unroll") for (int v = 0; v < V; ++v) {
          if (a.out_channel_major) a.out_t[((size_t)b * Co + c0 + v) * M + j] = best[v];
          else a.out_t[orow + v] = best[v];
          if (a.kstar_out) a.kstar_out[orow + v] = (unsigned char)kb[v];
        }
      }
      }  // qp
      if constexpr (MODE == PW_TRAIN) {
        if (++since_flush == kFlushTiles) flush();  // uniform: every thread of the block walks the same tiles
      }
    }

    if constexpr (NACC > 0) {  // the four waves' banks, in wave order
      if (since_flush > 0) flush();
      __syncthreads();
      for (int t = threadIdx.x; t < LVN; t += 256) {
        const double sum = ((red[t] + red[LVN + t]) + red[2 * LVN + t]) + red[3 * LVN + t];
        const int c = ch * L * V + t / NACC;
        if (c < Co) a.partial[((size_t)blockIdx.x * Co + c) * kPartialW + (t - (t / NACC) * NACC)] = sum;
      }
      __syncthreads();
    }
  }
}

// ---- the arg-max term of the backward pass.  ReLU + max route dz(j,c) to ONE slot per (query, channel),
// i.e. to support point tstar(j,c): a scatter with one target per element.  A workgroup owns (cloud, 4
// channels, a range of T support points), streams those channels' (dz, tstar) rows -- channel-major, so
// fully coalesced -- and accumulates in LDS in double (ds_add_f64: sums of a few dozen floats are exact in
// 53 bits unless their magnitudes span more than ~2^22, so the result does not depend on the order the
// adds arrive in).  Same structure as group_bwd_lds_kernel.
struct HitArgs {
  const float *dz_cm;   // [B,Co,M]
  const int *ts_cm;     // [B,Co,M]
  float *hit_cm;        // [B,Co,N]
  int B, N, M, Co, T;
};

__device__ __forceinline__ void hit_block(const HitArgs &a, int b, double *hacc /* [4][T] */) {
  const int c0 = blockIdx.x * 4;
  const int n0 = blockIdx.y * a.T;
  const int M = a.M, T = a.T;
  const int nch = a.Co - c0 < 4 ? a.Co - c0 : 4;
  const unsigned span = (unsigned)(a.N - n0 < T ? a.N - n0 : T);
  for (int t = threadIdx.x; t < 4 * T; t += 1024) hacc[t] = 0.0;
  __syncthreads();
  const float *dz = a.dz_cm + ((size_t)b * a.Co + c0) * M;
  const int *ts = a.ts_cm + ((size_t)b * a.Co + c0) * M;
  if ((M & 3) == 0) {
    for (int j = threadIdx.x * 4; j < M; j += 4096) {
      float4 d[4];
      int4 t[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const size_t o = (size_t)(v < nch ? v : 0) * M + j;
        d[v] = *reinterpret_cast<const float4 *>(dz + o);
        t[v] = *reinterpret_cast<const int4 *>(ts + o);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        if (v >= nch) break;
        double *hv = hacc + (size_t)v * T;
        if (d[v].x != 0.f && (unsigned)(t[v].x - n0) < span) atomicAdd(&hv[t[v].x - n0], (double)d[v].x);
        if (d[v].y != 0.f && (unsigned)(t[v].y - n0) < span) atomicAdd(&hv[t[v].y - n0], (double)d[v].y);
        if (d[v].z != 0.f && (unsigned)(t[v].z - n0) < span) atomicAdd(&hv[t[v].z - n0], (double)d[v].z);
        if (d[v].w != 0.f && (unsigned)(t[v].w - n0) < span) atomicAdd(&hv[t[v].w - n0], (double)d[v].w);
      }
    }
  } else {
    for (int j = threadIdx.x; j < M; j += 1024)
      for (int v = 0; v < nch; ++v) {
        const float d = dz[(size_t)v * M + j];
        const int t = ts[(size_t)v * M + j];
        if (d != 0.f && (unsigned)(t - n0) < span) atomicAdd(&hacc[(size_t)v * T + t - n0], (double)d);
      }
  }
  __syncthreads();
  for (int v = 0; v < nch; ++v)
    for (int i = threadIdx.x; i < (int)span; i += 1024)
      a.hit_cm[((size_t)b * a.Co + c0 + v) * a.N + n0 + i] = (float)hacc[(size_t)v * T + i];
}

__global__ __launch_bounds__(1024) void pwmlp_hit_kernel(HitArgs a) {
  extern __shared__ double hacc[];  // [4][T]
  hit_block(a, blockIdx.z, hacc);
}

// Few queries per cloud (the deep stages: M = 16 .. 1024 queries, hundreds of channels).  hit_block gives a block four
// channel rows and walks them 4096 queries at a time: at M = 16 four of its 1024 threads have work, and the 4608 blocks of a
// 1152-channel layer take 20 us for 1.2 MB.  Here a block owns CHB = 4096 / M channel rows -- thread t the four queries
// 4 (t % (M/4)) .. of row t / (M/4): one 16-byte load per thread and array -- of ONE cloud whose N <= 16384 / CHB support
// points all fit the block's LDS accumulators.  Same sums (doubles of a few dozen floats: exact, order-free).
__global__ __launch_bounds__(1024) void pwmlp_hit_wide_kernel(HitArgs a, int CHB) {
  extern __shared__ double hacc[];  // [CHB][N]
  const int b = blockIdx.z;
  const int c0 = blockIdx.x * CHB;
  const int M = a.M, N = a.N, q = M >> 2;
  const int nch = a.Co - c0 < CHB ? a.Co - c0 : CHB;
  for (int t = threadIdx.x; t < nch * N; t += 1024) hacc[t] = 0.0;
  __syncthreads();
  const int v = threadIdx.x / q, j = 4 * (threadIdx.x - v * q);
  if (v < nch) {
    const size_t o = ((size_t)b * a.Co + c0 + v) * M + j;
    const float4 d = *reinterpret_cast<const float4 *>(a.dz_cm + o);
    const int4 t = *reinterpret_cast<const int4 *>(a.ts_cm + o);
    double *hv = hacc + (size_t)v * N;
    if (d.x != 0.f && (unsigned)t.x < (unsigned)N) atomicAdd(&hv[t.x], (double)d.x);
    if (d.y != 0.f && (unsigned)t.y < (unsigned)N) atomicAdd(&hv[t.y], (double)d.y);
    if (d.z != 0.f && (unsigned)t.z < (unsigned)N) atomicAdd(&hv[t.z], (double)d.z);
    if (d.w != 0.f && (unsigned)t.w < (unsigned)N) atomicAdd(&hv[t.w], (double)d.w);
  }
  __syncthreads();
  float *out = a.hit_cm + ((size_t)b * a.Co + c0) * N;  // the block's rows are consecutive: one contiguous run
  for (int e = threadIdx.x; e < nch * N; e += 1024) out[e] = (float)hacc[e];
}

// channel rows per block of the wide form, 0 where it does not apply (M a power of two in 4 .. 1024 so that the threads
// tile CHB rows exactly; at least 8 rows -- below that hit_block's four rows per block do as well)
static int hit_wide_rows(int N, int M) {
  if (M < 4 || M > 1024 || (M & (M - 1)) != 0) return 0;
  int chb = 4096 / M;
  while (chb > 0 && (long long)chb * N > 16384) chb >>= 1;  // 128 KB of double accumulators
  return chb >= 8 ? chb : 0;
}

// support-major backward pass through the CSR inverse of idx.  For support point i with slot list S_i:
//   dG_i = sum_{s in S_i} dy_s,  dy_s = D y_s + Bc + A dz [slot s is the arg-max],  y_s = W_r rel_s + H[centre_s] + G_i
//        = D (W_r . sum rel_s + sum H[centre_s] + |S_i| G_i) + |S_i| Bc + A hit_i
// so a slot costs its relative position (rebuilt from the coordinates), its query's centre index and the H half of
// the centre's row of ght (a row gather, like the forward pass's G half); hit_i = the arg-max term from pwmlp_hit_kernel.
// dH_i = sum over the queries centred on i (= the slots (j,0) of S_i, reference :290) of
//        D sum_k y + K Bc + A dz  (sum_k y was left behind by the forward pass).
//
// Round 4 form (one kernel; rounds 2 / 3 had a slot-staging kernel, 74-88 us, then a per-geometry summary kernel on
// the index stream, 42 us, in front of a gather-only pass, 48 us).  A lane group (L lanes x V channels) owns one
// support point of a tile of 4 * QW points; persistent workgroups walk the tiles.  Per ENTRY of the point's list the
// work is done by ONE lane of the group -- lane cl takes entries cl, cl + L, ...: the slot id from inv_slots, its query
// j = slot / K, that query's 16-byte record {coordinates, centre idx[j, 0]} out of the table cl3d_pwmlp_bwd_rows leaves
// (one L1-friendly gather), the relative position with the forward pass's own expression -- and only the row gathers
// are per (entry, lane): the centre is handed round the group by shuffle and every lane reads its 16 bytes of the
// centre's H half-row, SB rows in flight.  A slot (j, 0) -- the query is centred on this very point -- gathers the
// point's own row (no branch in the loop) and adds the query's sum_k y / dz rows to dH_i.  The chain row bounds ->
// slot ids -> query records is three dependent round trips; it runs one, two and three tiles AHEAD of the rows being
// gathered (a workgroup has only a handful of tiles: unpipelined the chain was a quarter of the pass).
constexpr unsigned kCentreFlag = 0x80000000u;
constexpr int kSupRounds = 2;  // entries per lane whose records are fetched ahead (lists up to 2 L; the rest inline)

// sum over the L lanes of a lane group, returned to every lane of the group.  L == 16: the groups are the DPP rows
// (quad swaps, row mirrors: no LDS).  Other widths go through a per-wave LDS scratch.
__device__ __forceinline__ float group_sum(float v, int L, int lane, int first /* first lane of the group, -1: none */,
                                           float *scratch /* [64] of this wave */) {
  if (L == 16) {
#define CL3D_DPP_ADD(ctrl) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    CL3D_DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    CL3D_DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    CL3D_DPP_ADD(0x141);  // row_half_mirror
    CL3D_DPP_ADD(0x140);  // row_mirror: every lane holds its row's 16-lane sum
#undef CL3D_DPP_ADD
    return v;
  }
  scratch[lane] = v;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float t = 0.f;
  if (first >= 0)
    for (int l = 0; l < L; ++l) t += scratch[first + l];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return t;
}

#ifndef CL3D_SUP_SB
#define CL3D_SUP_SB 4      // H rows in flight per lane   } tunables of the support-major pass: the defaults ship,
#endif                     //                             } scripts/micro/kernel_variants.py times the others
#ifndef CL3D_SUP_WAVES
#define CL3D_SUP_WAVES 4   // waves per SIMD the register budget is set for
#endif
#ifndef CL3D_SUP_LATE
#define CL3D_SUP_LATE 0    // 1: the tile's arg-max block and the point's own G row are requested after the gather loop
#endif                     //    (eight registers less across it, one more exposed round trip per tile)
template <int V, int SB>
__global__ __launch_bounds__(256, CL3D_SUP_WAVES) void pwmlp_support_kernel(PwArgs a) {
  __shared__ float s_hit[1280];   // [L * V channels][TR + 1]: L * V * (256 / L + 1) <= 4 * (256 + 64) floats
  __shared__ float s_con[6][256]; // wr (3), A, Bc, D of the chunk's L * V <= 256 channels
  __shared__ float s_grp[4][64];  // group_sum scratch (lane groups that are not DPP rows)
  const int K = a.K, Co = a.Co, M = a.M, N = a.N, L = a.L, QW = a.QW;
  const int row = 2 * Co;
  const unsigned rowb = (unsigned)row * 4u;
  const int MK = M * K;
  const int TR = 4 * QW, LV = L * V;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int g = lane / L, cl = lane - g * L;
  const int tiles_per_cloud = (N + TR - 1) / TR;
  const int ntiles = a.B * tiles_per_cloud;
  const bool grp_on = g < QW;  // lanes past the last whole group (64 % L) only keep the barriers
  // A tile is TR consecutive points of ONE cloud: the cloud is workgroup-uniform (kept on the scalar side), the point
  // belongs to the lane group.
  struct Pt {  // this lane group's point of a tile: row (N: none), list start and length
    int i, s0, len;
  };
  struct En {  // this lane's entry of a round: byte offset of the row the entry adds in the cloud's ght (bit 31: the
    unsigned off, j;  // entry is a slot (j, 0) -- the row is this point's own, j the query whose rows feed d H)
  };
  auto tile_cloud = [&](int t, int &tr) {
    int b = 0;
    tr = 0;
    if (t < ntiles) decode_tile(t, a.B, tiles_per_cloud, b, tr);
    return __builtin_amdgcn_readfirstlane(b);
  };
  auto fetch_point = [&](int t, int b, int tr) {
    Pt p{N, 0, 0};
    if (t < ntiles) {
      const int i = tr * TR + wave * QW + g;
      if (grp_on && i < N) {
        const int *off = a.inv_off + (size_t)b * (N + 1) + i;
        p.i = i;
        p.s0 = off[0];
        p.len = off[1] - p.s0;
      }
    }
    return p;
  };
  auto fetch_slot_at = [&](const Pt &p, int b, int e) {  // slot id of the list's entry e, -1 past its end
    return e < p.len ? a.inv_slots[(size_t)b * MK + p.s0 + e] : -1;
  };
  // the entry of a slot, and its relative position added to (rx, ry, rz) -- the forward pass's own expression for rel
  // (pwmlp_query_kernel's slot record)
  auto fetch_entry = [&](const Pt &p, int b, int slot, float &rx, float &ry, float &rz) {
    En e{0u, 0u};
    if (slot >= 0) {
      const int j = div_k(slot, a.kmagic, K);
      const float4 qt = a.qtab[(size_t)b * M + j];  // {query coordinates, centre idx[j, 0]}
      const float *pp = a.support_xyz + ((size_t)b * N + p.i) * 3;
      rx += (pp[0] - qt.x) * a.inv_radius;
      ry += (pp[1] - qt.y) * a.inv_radius;
      rz += (pp[2] - qt.z) * a.inv_radius;
      const bool centred = slot - j * K == 0;
      e.off = (centred ? (unsigned)p.i : (unsigned)__float_as_int(qt.w)) * rowb | (centred ? kCentreFlag : 0u);
      e.j = (unsigned)j;
    }
    return e;
  };
  for (int ch = blockIdx.y; ch < a.chunks; ch += gridDim.y) {
    // a lane whose channels lie past Co (last chunk) still carries entries for its group: it reads channel 0's
    // pieces and stores nothing
    const bool chan_on = (ch * L + cl) * V < Co;
    const int c0 = chan_on ? (ch * L + cl) * V : 0;
    const int cbase = ch * LV;
    __syncthreads();  // the previous chunk's readers are done with the constants
    for (int t = threadIdx.x; t < LV; t += 256) {
      const int c = cbase + t < Co ? cbase + t : Co - 1;
      s_con[0][t] = a.wr[c * 3 + 0]; s_con[1][t] = a.wr[c * 3 + 1]; s_con[2][t] = a.wr[c * 3 + 2];
      s_con[3][t] = a.v0[c]; s_con[4][t] = a.v1[c]; s_con[5][t] = a.v2[c];
    }
    // ---- the look-ahead chain: point (tile + 3 G) -> slots (tile + 2 G) -> entries (tile + G) -> rows (tile)
    int tile = blockIdx.x;
    const int G = (int)gridDim.x;
    int tr0, tr1, tr2;
    int b0 = tile_cloud(tile, tr0), b1 = tile_cloud(tile + G, tr1), b2 = tile_cloud(tile + 2 * G, tr2);
    Pt p0 = fetch_point(tile, b0, tr0), p1 = fetch_point(tile + G, b1, tr1), p2 = fetch_point(tile + 2 * G, b2, tr2);
    int sl1[kSupRounds];
    En e0[kSupRounds];
    float rs0[3] = {0.f, 0.f, 0.f};  // this lane's share of the current point's sum rel
#pragma unroll
    for (int r = 0; r < kSupRounds; ++r) e0[r] = fetch_entry(p0, b0, fetch_slot_at(p0, b0, r * L + cl), rs0[0], rs0[1], rs0[2]);
#pragma unroll
    for (int r = 0; r < kSupRounds; ++r) sl1[r] = fetch_slot_at(p1, b1, r * L + cl);
    for (; tile < ntiles; tile += G) {
      const bool row_on = p0.i < N;
      const int i0 = tr0 * TR;
      // --- requests for later tiles
      En e1[kSupRounds];
      float rs1[3] = {0.f, 0.f, 0.f};
      int sl2[kSupRounds];
#pragma unroll
      for (int r = 0; r < kSupRounds; ++r) e1[r] = fetch_entry(p1, b1, sl1[r], rs1[0], rs1[1], rs1[2]);
#pragma unroll
      for (int r = 0; r < kSupRounds; ++r) sl2[r] = fetch_slot_at(p2, b2, r * L + cl);
      int tr3;
      const int b3 = tile_cloud(tile + 3 * G, tr3);
      const Pt p3 = fetch_point(tile + 3 * G, b3, tr3);
      // --- this tile: arg-max terms (to LDS below), the point's own row
      float4 h4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool hit_vec = (N & 3) == 0 && LV * (TR / 4) <= 256;
      auto load_hits = [&]() {
        if (hit_vec) {
          const int q4 = TR / 4, cc = threadIdx.x / q4, qd = threadIdx.x - cc * q4;
          if (cc < LV && cbase + cc < Co && i0 + qd * 4 < N)
            h4 = *reinterpret_cast<const float4 *>(a.hit_cm + ((size_t)b0 * Co + cbase + cc) * N + i0 + qd * 4);
        }
      };
      if (!CL3D_SUP_LATE) load_hits();
      float shc[V], csy[V], cdz[V];
      int ncen = 0;  // queries centred on this point (flagged entries)
      Vec<V> gi;
#pragma unroll
      for (int v = 0; v < V; ++v) shc[v] = csy[v] = cdz[v] = gi.v[v] = 0.f;
      if (row_on) {
        const int len = p0.len;
        // uniform bases + 32-bit lane offsets: the gathers are saddr + voffset loads (one VGPR per address)
        const char *hrows = reinterpret_cast<const char *>(a.ght + (size_t)b0 * N * row) + ((size_t)Co + (size_t)c0) * 4u;
        const char *syrows = reinterpret_cast<const char *>(a.sy_in + (size_t)b0 * M * Co) + (size_t)c0 * 4u;
        const char *dzrows = reinterpret_cast<const char *>(a.dz_t + (size_t)b0 * M * Co) + (size_t)c0 * 4u;
        if (!CL3D_SUP_LATE) gi = load_row<V>(reinterpret_cast<const float *>(hrows + (size_t)p0.i * rowb) - Co);
        const unsigned long long gmask = L >= 64 ? ~0ull : ((1ull << L) - 1ull);
        auto round_of = [&](const En &en, int p) {  // entries p .. p + L - 1 of the list, one per lane of the group
          const int nr = len - p < L ? len - p : L;
          // the round's flagged entries (queries centred on this point), as a bit mask over the group's lanes; the
          // first one's two query-major rows are requested now, ahead of the round's H rows
          unsigned long long fm = (__ballot(p + cl < len && (en.off >> 31) != 0u) >> (g * L)) & gmask;
          ncen += (int)__popcll(fm);
          Vec<V> ry0, rd0;
#pragma unroll
          for (int v = 0; v < V; ++v) ry0.v[v] = rd0.v[v] = 0.f;
          if (fm != 0ull) {  // (uniform within the lane group)
            const int l = __ffsll((long long)fm) - 1;
            fm &= fm - 1ull;
            const unsigned o = (unsigned)__shfl((int)en.j, g * L + l, CL3D_WAVE) * ((unsigned)Co * 4u);
            ry0 = load_row<V>(reinterpret_cast<const float *>(syrows + o));
            rd0 = load_row<V>(reinterpret_cast<const float *>(dzrows + o));
          }
          int u0 = 0;
          for (; u0 + SB <= nr; u0 += SB) {  // full batches: no per-entry guard
            Vec<V> rr[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
              const unsigned o = (unsigned)__shfl((int)en.off, g * L + u0 + u, CL3D_WAVE) & 0x7fffffffu;
              rr[u] = load_row<V>(reinterpret_cast<const float *>(hrows + o));
            }
#pragma unroll
            for (int u = 0; u < SB; ++u)
#pragma unroll
              for (int v = 0; v < V; ++v) shc[v] += rr[u].v[v];
          }
          if (u0 < nr) {  // the list's last, partial batch
            Vec<V> rr[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
              const unsigned o = (unsigned)__shfl((int)en.off, g * L + (u0 + u < nr ? u0 + u : nr - 1), CL3D_WAVE) & 0x7fffffffu;
              rr[u] = load_row<V>(reinterpret_cast<const float *>(hrows + o));
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
              if (u0 + u >= nr) continue;
#pragma unroll
              for (int v = 0; v < V; ++v) shc[v] += rr[u].v[v];
            }
          }
#pragma unroll
          for (int v = 0; v < V; ++v) {
            csy[v] += ry0.v[v];
            cdz[v] += rd0.v[v];
          }
          while (fm != 0ull) {  // further centred queries of the round (duplicated points): one at a time
            const int l = __ffsll((long long)fm) - 1;
            fm &= fm - 1ull;
            const unsigned o = (unsigned)__shfl((int)en.j, g * L + l, CL3D_WAVE) * ((unsigned)Co * 4u);
            const Vec<V> ry = load_row<V>(reinterpret_cast<const float *>(syrows + o));
            const Vec<V> rd = load_row<V>(reinterpret_cast<const float *>(dzrows + o));
#pragma unroll
            for (int v = 0; v < V; ++v) {
              csy[v] += ry.v[v];
              cdz[v] += rd.v[v];
            }
          }
        };
        int p = 0;
#pragma unroll
        for (int r = 0; r < kSupRounds; ++r) {  // the rounds whose records came with the look-ahead
          if (p < len) {
            round_of(e0[r], p);
            p += L;
          }
        }
        for (; p < len; p += L)  // long lists: fetched in line
          round_of(fetch_entry(p0, b0, fetch_slot_at(p0, b0, p + cl), rs0[0], rs0[1], rs0[2]), p);
        if (CL3D_SUP_LATE) gi = load_row<V>(reinterpret_cast<const float *>(hrows + (size_t)p0.i * rowb) - Co);
      }
      if (CL3D_SUP_LATE) load_hits();
      // sum rel over the group's lanes (every lane of the wave takes part; lanes without a row carry zeros)
      const float rsx = group_sum(rs0[0], L, lane, grp_on ? g * L : -1, s_grp[wave]);
      const float rsy = group_sum(rs0[1], L, lane, grp_on ? g * L : -1, s_grp[wave]);
      const float rsz = group_sum(rs0[2], L, lane, grp_on ? g * L : -1, s_grp[wave]);
      __syncthreads();  // the previous tile's readers are done with s_hit (and the constants are written)
      if (hit_vec) {
        const int q4 = TR / 4, cc = threadIdx.x / q4, qd = threadIdx.x - cc * q4;
        if (cc < LV) {
          float *d = s_hit + cc * (TR + 1) + qd * 4;
          d[0] = h4.x; d[1] = h4.y; d[2] = h4.z; d[3] = h4.w;
        }
      } else {
        for (int t = threadIdx.x; t < LV * TR; t += 256) {
          const int cc = t / TR, ii = t - cc * TR;
          if (cbase + cc < Co && i0 + ii < N) s_hit[cc * (TR + 1) + ii] = a.hit_cm[((size_t)b0 * Co + cbase + cc) * N + i0 + ii];
        }
      }
      __syncthreads();
      if (row_on && chan_on) {
        const float cnt = (float)p0.len, fcen = (float)ncen;
        float *dst = a.dght + ((size_t)b0 * N + p0.i) * row + c0;
        Vec<V> dg, dh;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const int cc = cl * V + v;
          const float hit = s_hit[cc * (TR + 1) + (p0.i - i0)];
          float t = s_con[0][cc] * rsx;
          t = __builtin_fmaf(s_con[1][cc], rsy, t);
          t = __builtin_fmaf(s_con[2][cc], rsz, t);
          const float cA = s_con[3][cc], cB = s_con[4][cc], cD = s_con[5][cc];
          const float ysum = (t + shc[v]) + cnt * gi.v[v];  // (shc holds this point's own H row once per centred query)
          dg.v[v] = __builtin_fmaf(cD, ysum, __builtin_fmaf(cA, hit, cnt * cB));
          dh.v[v] = __builtin_fmaf(cD, csy[v], __builtin_fmaf(cA, cdz[v], fcen * ((float)K * cB)));
        }
        store_row<V>(dst, dg);
        store_row<V>(dst + Co, dh);
      }
      p0 = p1; p1 = p2; p2 = p3;
      b0 = b1; b1 = b2; b2 = b3;
      tr0 = tr1; tr1 = tr2; tr2 = tr3;
#pragma unroll
      for (int r = 0; r < kSupRounds; ++r) {
        e0[r] = e1[r];
        sl1[r] = sl2[r];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) rs0[c] = rs1[c];
    }
  }
}

// ---- element-wise passes over the per-(query, channel) rows ------------------------------------------
// Tile = 64 queries x CW channels (CW a power of two <= 64: Co is walked in its binary decomposition, so a
// thread keeps ONE channel for the whole pass and its partial sums stay in registers).  The channel-major
// side of the transposition ([B,Co,M] output / upstream gradient) goes through an LDS tile, so both sides
// are read and written in full 256-byte rows.
enum { ROWS_APPLY = 0, ROWS_BWD = 1 };
constexpr int kRowsBatch = 4;

struct RowArgs {
  const float *ystar_t;           // [B,M,Co]
  const unsigned char *kstar_t;   // [B,M,Co]
  const int *idx;                 // [B,M,K]: the arg-max slot's support index is idx[j, kstar]
  const float *query_xyz, *support_xyz;  // BWD: rel of the arg-max slot is rebuilt from them
  float inv_radius;
  int N;
  const float *gout;              // [B,Co,M] (channel-major) or [B,M,Co]
  int gout_channel_major;
  const float *scale, *shift, *mean, *invstd;
  float *out;                     // APPLY: [B,Co,M]
  float *dz_cm;                   // BWD: [B,Co,M] gated upstream gradient
  float *dz_t;                    // BWD: [B,M,Co] the same, point-major
  float4 *qtab;                   // BWD: [B,M] {query coordinates, idx[j, 0]} for the support-major pass
  int *ts_cm;                     // BWD: [B,Co,M] tstar, transposed for the hit pass
  double *partial;                // BWD: [gridDim.x, Co, kPartialW]
  int B, M, K, Co;
};

template <int MODE>
__global__ __launch_bounds__(256) void pwmlp_rows_kernel(RowArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[64 * 65];
  __shared__ int tile2[MODE == ROWS_BWD ? 64 * 65 : 1];
  // the end-of-kernel reduction reuses the tile (every reader of it has passed the loop's closing barrier): 33 KB per
  // workgroup instead of 43, four workgroups per CU instead of three -- the BWD pass's 1024 one-tile workgroups are then
  // all resident at once
  double *red = reinterpret_cast<double *>(tile);
  static_assert(sizeof(tile) >= 256 * 5 * sizeof(double), "reduction scratch must fit the tile");
  const int M = a.M, Co = a.Co, K = a.K;
  const int tid = threadIdx.x;
  const int tiles_per_cloud = (M + 63) / 64;
  const int ntiles = a.B * tiles_per_cloud;
  // channel chunks: Co / 64 full ones, then the binary decomposition of the remainder; spread over gridDim.y
  const int nfull = Co >> 6, nchunks = nfull + __builtin_popcount(Co & 63);
  for (int q = blockIdx.y; q < nchunks; q += gridDim.y) {
    int cbase = (q < nfull ? q : nfull) * 64, CW = 64;
    for (int r = q - nfull; r >= 0; --r) {
      while (CW > Co - cbase) CW >>= 1;
      if (r > 0) cbase += CW;
    }
    const int cl = tid & (CW - 1), r0 = tid / CW, RS = 256 / CW;
    const int c = cbase + cl;
    const float scale = a.scale[c], shift = a.shift[c];
    float mean = 0.f, invstd = 0.f;
    if constexpr (MODE == ROWS_BWD) {
      mean = a.mean[c];
      invstd = a.invstd[c];
    }
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const int b = t / tiles_per_cloud;
      const int j0 = (t - b * tiles_per_cloud) * 64;
      const int nj = M - j0 < 64 ? M - j0 : 64;
      if constexpr (MODE == ROWS_BWD) {
        if (a.gout_channel_major) {
          for (int e0 = 0; e0 < CW * 64; e0 += 256 * kRowsBatch) {
            float gv[kRowsBatch];
#pragma unroll
            for (int u = 0; u < kRowsBatch; ++u) {
              const int e = e0 + u * 256 + tid;
              const int cc = (e >> 6) < CW ? (e >> 6) : CW - 1, jq = e & 63;
              gv[u] = a.gout[((size_t)b * Co + cbase + cc) * M + j0 + (jq < nj ? jq : nj - 1)];
            }
#pragma unroll
            for (int u = 0; u < kRowsBatch; ++u) {
              const int e = e0 + u * 256 + tid;
              if ((e >> 6) < CW) tile[(e >> 6) * 65 + (e & 63)] = gv[u];
            }
          }
          __syncthreads();
        }
        for (int jb = r0; jb < nj; jb += RS * kRowsBatch) {
          float y[kRowsBatch], gq[kRowsBatch];
          int ks[kRowsBatch], ts[kRowsBatch];
          float4 rel[kRowsBatch];
#pragma unroll
          for (int u = 0; u < kRowsBatch; ++u) {
            const int jq = jb + u * RS < nj ? jb + u * RS : nj - 1;
            const size_t e = ((size_t)b * M + j0 + jq) * Co + c;
            y[u] = a.ystar_t[e];
            ks[u] = a.kstar_t[e];
            gq[u] = a.gout_channel_major ? tile[cl * 65 + jq] : a.gout[e];
          }
          if (CW == 64 && K <= 64) {
            // a wave = the 64 channels of ONE query: lane l fetches slot l's neighbour index and coordinates -- loads
            // that do not wait for the arg-max slots -- and every lane then picks its own slot's by shuffle (two
            // dependent round trips per batch instead of three)
            int iv[kRowsBatch];
            float sx[kRowsBatch], sy[kRowsBatch], sz[kRowsBatch], qx[kRowsBatch], qy[kRowsBatch], qz[kRowsBatch];
#pragma unroll
            for (int u = 0; u < kRowsBatch; ++u) {
              const int jq = jb + u * RS < nj ? jb + u * RS : nj - 1;
              iv[u] = a.idx[((size_t)b * M + j0 + jq) * K + (cl < K ? cl : K - 1)];
              const float *qp = a.query_xyz + ((size_t)b * M + j0 + jq) * 3;
              qx[u] = qp[0]; qy[u] = qp[1]; qz[u] = qp[2];
            }
#pragma unroll
            for (int u = 0; u < kRowsBatch; ++u) {
              const float *sp = a.support_xyz + ((size_t)b * a.N + iv[u]) * 3;
              sx[u] = sp[0]; sy[u] = sp[1]; sz[u] = sp[2];
            }
#pragma unroll
            for (int u = 0; u < kRowsBatch; ++u) {
              ts[u] = __shfl(iv[u], ks[u], CL3D_WAVE);
              rel[u] = make_float4((__shfl(sx[u], ks[u], CL3D_WAVE) - qx[u]) * a.inv_radius,
                                   (__shfl(sy[u], ks[u], CL3D_WAVE) - qy[u]) * a.inv_radius,
                                   (__shfl(sz[u], ks[u], CL3D_WAVE) - qz[u]) * a.inv_radius, 0.f);
            }
          } else {
#pragma unroll
          for (int u = 0; u < kRowsBatch; ++u) {
            const int jq = jb + u * RS < nj ? jb + u * RS : nj - 1;
            ts[u] = a.idx[((size_t)b * M + j0 + jq) * K + ks[u]];  // a wave = the channels of one query: one 128-byte row of idx
          }
#pragma unroll
          for (int u = 0; u < kRowsBatch; ++u) {  // rel of that slot, with the forward pass's expression
            const int jq = jb + u * RS < nj ? jb + u * RS : nj - 1;
            const float *sp = a.support_xyz + ((size_t)b * a.N + ts[u]) * 3;
            const float *qp = a.query_xyz + ((size_t)b * M + j0 + jq) * 3;
            rel[u] = make_float4((sp[0] - qp[0]) * a.inv_radius, (sp[1] - qp[1]) * a.inv_radius,
                                 (sp[2] - qp[2]) * a.inv_radius, 0.f);
          }
          }
#pragma unroll
          for (int u = 0; u < kRowsBatch; ++u) {
            const int jq = jb + u * RS;
            if (jq >= nj) continue;
            const float z = __builtin_fmaf(y[u], scale, shift);
            const float dz = z > 0.f ? gq[u] : 0.f;
            tile[cl * 65 + jq] = dz;  // own element of the tile: overwritten in place, transposed out below
            tile2[cl * 65 + jq] = ts[u];
            // ... and a point-major copy: the support-major pass reads the rows of the queries centred on a point
            // (one 16-byte piece per lane; out of the channel-major array that was 64 L2 requests per centred query)
            a.dz_t[((size_t)b * M + j0 + jq) * Co + c] = dz;
            acc[0] += (double)dz;
            acc[1] += (double)(dz * ((y[u] - mean) * invstd));
            acc[2] += (double)(dz * rel[u].x);
            acc[3] += (double)(dz * rel[u].y);
            acc[4] += (double)(dz * rel[u].z);
          }
        }
        if (a.qtab != nullptr && q == 0 && tid < nj) {  // the query table of the slot-walking support pass: {coordinates, centre idx[j, 0]}
          const size_t j = (size_t)b * M + j0 + tid;
          const float *qp = a.query_xyz + j * 3;
          a.qtab[j] = make_float4(qp[0], qp[1], qp[2], __int_as_float(a.idx[j * K]));
        }
        __syncthreads();
        for (int e = tid; e < CW * 64; e += 256) {  // dz and its target, channel-major, for the hit pass
          const int cc = e >> 6, jq = e & 63;
          if (jq < nj) {
            const size_t o = ((size_t)b * Co + cbase + cc) * M + j0 + jq;
            a.dz_cm[o] = tile[cc * 65 + jq];
            a.ts_cm[o] = tile2[cc * 65 + jq];
          }
        }
        __syncthreads();  // the tile is overwritten by the next iteration
      } else {
        for (int jb = r0; jb < nj; jb += RS * kRowsBatch) {
          float y[kRowsBatch];
#pragma unroll
          for (int u = 0; u < kRowsBatch; ++u) {
            const int jq = jb + u * RS < nj ? jb + u * RS : nj - 1;
            y[u] = a.ystar_t[((size_t)b * M + j0 + jq) * Co + c];
          }
#pragma unroll
          for (int u = 0; u < kRowsBatch; ++u) {
            const int jq = jb + u * RS;
            if (jq >= nj) continue;
            const float z = __builtin_fmaf(y[u], scale, shift);
            tile[cl * 65 + jq] = z > 0.f ? z : 0.f;
          }
        }
        __syncthreads();
        for (int e = tid; e < CW * 64; e += 256) {
          const int cc = e >> 6, jq = e & 63;
          if (jq < nj) a.out[((size_t)b * Co + cbase + cc) * M + j0 + jq] = tile[cc * 65 + jq];
        }
        __syncthreads();
      }
    }
    if constexpr (MODE == ROWS_BWD) {  // fixed-order reduction over the RS threads that share a channel
#pragma unroll
      for (int p = 0; p < 5; ++p) red[tid * 5 + p] = acc[p];
      __syncthreads();
      for (int e = tid; e < CW * 5; e += 256) {
        const int cc = e / 5, p = e - cc * 5;
        double sum = 0.0;
        for (int r = 0; r < RS; ++r) sum += red[(r * CW + cc) * 5 + p];
        a.partial[((size_t)blockIdx.x * Co + cbase + cc) * kPartialW + p] = sum;
      }
      __syncthreads();
    }
  }
}

// ---- the same two passes for whole 64-channel chunks and whole 64-query tiles (every layer whose width is a
// multiple of 64 on clouds of a multiple of 64 points: the metric shape), with ALL of a tile's loads in flight at once.
// pwmlp_rows_kernel walks a tile in four batches, each a chain of dependent round trips (gout tile -> barrier -> per
// batch: y*, k*, idx row -> the slot's coordinates): with one tile per workgroup and every workgroup resident the
// whole launch is ONE such chain, ~10 round trips = 32 us for 82 MB at the metric shape.  Here a wave owns 16
// consecutive queries: the upstream-gradient tile (16-byte loads), y*, k*, the 16 idx rows and the queries' own
// coordinates (wave-uniform: scalar loads) are requested before anything is waited for, the arg-max slots' support
// coordinates follow in two halves (the second half's gathers fly during the first half's arithmetic), and the
// channel-major results leave as 16-byte stores.  Same expressions per element as pwmlp_rows_kernel; the double
// partials are summed in (wave, query) order -- a fixed order, but not pwmlp_rows_kernel's.
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
template <typename T>
__device__ __forceinline__ T ld_at(const char *base, unsigned byte_off) {  // uniform base + 32-bit lane offset: one access
  return *reinterpret_cast<const T *>(base + byte_off);
}

template <int MODE>
__global__ __launch_bounds__(256, 4) void pwmlp_rows64_kernel(RowArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[64 * 65];
  __shared__ int tile2[MODE == ROWS_BWD ? 64 * 65 : 1];
  double *red = reinterpret_cast<double *>(tile);
  const int M = a.M, Co = a.Co, K = a.K;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles_per_cloud = M >> 6;
  const int ntiles = a.B * tiles_per_cloud;
  const int cbase = blockIdx.y * 64;  // one 64-channel chunk per blockIdx.y (the launch sets gridDim.y = Co / 64)
  const unsigned K4 = 4u * (unsigned)K, Co4 = 4u * (unsigned)Co, M4 = 4u * (unsigned)M;
  double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    // Every lane-dependent offset below is rebuilt per tile from an opaque copy of the thread index: as loop invariants
    // they were hoisted out of the tile loop (zero-extended to 64-bit pairs, one per access) and spilled -- and a kernel
    // with a scratch frame is not allowed next to another one on the captured step's queues.
    unsigned tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const unsigned cl = tid & 63u;
    const unsigned cc4 = tid >> 4, j4 = (tid & 15u) * 4u;  // channel-major side: channel rows cc4 + 16 u, four queries
    const unsigned g4 = cc4 * M4 + 4u * j4;
    const unsigned c4 = 4u * ((unsigned)cbase + cl);
    const float scale = ld_at<float>(reinterpret_cast<const char *>(a.scale), c4);
    const float shift = ld_at<float>(reinterpret_cast<const char *>(a.shift), c4);
    const int b = t / tiles_per_cloud;
    const int j0 = (t - b * tiles_per_cloud) * 64;
    const size_t row0 = (size_t)b * M + j0 + 16 * w;  // first of this wave's 16 queries
    float *const trow = tile + cc4 * 65 + j4;         // + 16 u rows
    float *const town = tile + cl * 65 + 16 * w;      // + i: this lane's (channel, query) elements
    if constexpr (MODE == ROWS_BWD) {
      const float mean = ld_at<float>(reinterpret_cast<const char *>(a.mean), c4);
      const float invstd = ld_at<float>(reinterpret_cast<const char *>(a.invstd), c4);
      const char *gbase = reinterpret_cast<const char *>(a.gout + ((size_t)b * Co + cbase) * M + j0);
      const char *ibase = reinterpret_cast<const char *>(a.idx + row0 * K);
      const char *ybase = reinterpret_cast<const char *>(a.ystar_t + row0 * Co + cbase);
      const char *kbase = reinterpret_cast<const char *>(a.kstar_t + row0 * Co + cbase);
      const char *sbase = reinterpret_cast<const char *>(a.support_xyz + (size_t)b * a.N * 3);
      char *dbase = reinterpret_cast<char *>(a.dz_t + row0 * Co + cbase);
      char *zbase = reinterpret_cast<char *>(a.dz_cm + ((size_t)b * Co + cbase) * M + j0);
      char *tbase = reinterpret_cast<char *>(a.ts_cm + ((size_t)b * Co + cbase) * M + j0);
      // (1) everything that waits for nothing: the upstream-gradient tile, the wave's 16 idx rows, y*, the queries'
      // coordinates.  One running offset register per stream, bumped between the loads (sixteen precomputed offsets per
      // stream overflow the register file while the loads are in flight).
      float4 gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) gv[u] = ld_at<float4>(gbase, g4 + 16u * (unsigned)u * M4);
      int iv[16], ks[16];
      float y[16];
      unsigned off = 4u * (cl < (unsigned)K ? cl : (unsigned)K - 1u);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        iv[i] = ld_at<int>(ibase, off);
        off += K4;
        asm volatile("" : "+v"(off));
      }
      off = 4u * cl;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        y[i] = ld_at<float>(ybase, off);
        off += Co4;
        asm volatile("" : "+v"(off));
      }
      // the wave's 16 queries' coordinates: 48 consecutive floats, one per lane, handed out by v_readlane below
      const float qv = ld_at<float>(reinterpret_cast<const char *>(a.query_xyz + row0 * 3), 4u * (cl < 48u ? cl : 47u));
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // the gradient tile arrives first (requested first)
        float *row = trow + 16 * u * 65;
        row[0] = gv[u].x; row[1] = gv[u].y; row[2] = gv[u].z; row[3] = gv[u].w;
      }
      // (2) the arg-max slots, requested once the gradient tile's registers are free: first needed after the coordinate
      // gathers below, whose round trip they share
      off = cl;
      asm volatile("" : "+v"(off) : : "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        ks[i] = ld_at<unsigned char>(kbase, off);
        off += (unsigned)Co;
        asm volatile("" : "+v"(off));
      }
      // (3) lane l holds slot l's support index: its coordinates, for the first eight queries
      float sx[8], sy[8], sz[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float *sp = reinterpret_cast<const float *>(sbase + 12u * (unsigned)iv[i]);
        sx[i] = sp[0]; sy[i] = sp[1]; sz[i] = sp[2];
      }
      // the lane's arg-max slot rides in the top byte of its index word from here on (N < 2^24, K <= 64: host check)
#pragma unroll
      for (int i = 0; i < 16; ++i) iv[i] |= ks[i] << 24;
      __syncthreads();
      // the query table of the support-major pass, {coordinates, centre idx[j, 0]}: lane 0 holds both (its index word
      // is slot 0 of the query's row)
      const bool writes_qtab = a.qtab != nullptr && cbase == 0 && cl == 0;
      off = 4u * cl;
      asm volatile("" : "+v"(off));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float *sp = reinterpret_cast<const float *>(sbase + 12u * ((unsigned)iv[8 + i] & 0xffffffu));
            sx[i] = sp[0]; sy[i] = sp[1]; sz[i] = sp[2];
          }
        }
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
          const int i = 8 * h + ii;
          float qvi = qv;
          asm volatile("" : "+v"(qvi));  // keeps the 48 v_readlane results from being formed (and held in SGPRs) up front
          const float qx = lane_value(qvi, 3 * i), qy = lane_value(qvi, 3 * i + 1), qz = lane_value(qvi, 3 * i + 2);
          if (writes_qtab) a.qtab[row0 + i] = make_float4(qx, qy, qz, __int_as_float(iv[i] & 0xffffff));
          const int kq = (int)((unsigned)iv[i] >> 24);
          const int ts = __shfl(iv[i], kq, CL3D_WAVE) & 0xffffff;
          const float rx = (__shfl(sx[ii], kq, CL3D_WAVE) - qx) * a.inv_radius;
          const float ry = (__shfl(sy[ii], kq, CL3D_WAVE) - qy) * a.inv_radius;
          const float rz = (__shfl(sz[ii], kq, CL3D_WAVE) - qz) * a.inv_radius;
          const float z = __builtin_fmaf(y[i], scale, shift);
          const float dz = z > 0.f ? town[i] : 0.f;
          town[i] = dz;  // own element of the tile: overwritten in place, transposed out below
          (tile2 + (town - tile))[i] = ts;
          *reinterpret_cast<float *>(dbase + off) = dz;
          off += Co4;
          asm volatile("" : "+v"(off));
          acc[0] += (double)dz;
          acc[1] += (double)(dz * ((y[i] - mean) * invstd));
          acc[2] += (double)(dz * rx);
          acc[3] += (double)(dz * ry);
          acc[4] += (double)(dz * rz);
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float *row = trow + 16 * u * 65;
        const int *row2 = tile2 + (row - tile);
        const unsigned o = g4 + 16u * (unsigned)u * M4;  // the byte offsets the upstream-gradient tile was read at
        *reinterpret_cast<float4 *>(zbase + o) = make_float4(row[0], row[1], row[2], row[3]);
        *reinterpret_cast<int4 *>(tbase + o) = make_int4(row2[0], row2[1], row2[2], row2[3]);
      }
      __syncthreads();  // the tile is overwritten by the next iteration (and by the closing reduction)
    } else {
      const char *ybase = reinterpret_cast<const char *>(a.ystar_t + row0 * Co + cbase);
      char *obase = reinterpret_cast<char *>(a.out + ((size_t)b * Co + cbase) * M + j0);
      float y[16];
      unsigned off = 4u * cl;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        y[i] = ld_at<float>(ybase, off);
        off += Co4;
        asm volatile("" : "+v"(off));
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float z = __builtin_fmaf(y[i], scale, shift);
        town[i] = z > 0.f ? z : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float *row = trow + 16 * u * 65;
        *reinterpret_cast<float4 *>(obase + (g4 + 16u * (unsigned)u * M4)) = make_float4(row[0], row[1], row[2], row[3]);
      }
      __syncthreads();
    }
  }
  if constexpr (MODE == ROWS_BWD) {  // fixed-order reduction over the four waves that share a channel
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 5; ++p) red[tid * 5 + p] = acc[p];
    __syncthreads();
    for (int e = tid; e < 64 * 5; e += 256) {
      const int cc = e / 5, p = e - cc * 5;
      double sum = 0.0;
      for (int r = 0; r < 4; ++r) sum += red[(r * 64 + cc) * 5 + p];
      a.partial[((size_t)blockIdx.x * Co + cbase + cc) * kPartialW + p] = sum;
    }
  }
}

// whether the whole-tile form covers a rows pass: 64-channel chunks, 64-query tiles, 16-byte channel-major rows
static bool rows64_covers(const RowArgs &a, bool bwd) {
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  if (a.Co % 64 != 0 || a.M % 64 != 0) return false;
  if (!bwd) return al16(a.out);
  return a.K <= 64 && a.N < (1 << 24) && a.gout_channel_major && al16(a.gout) && al16(a.dz_cm) && al16(a.ts_cm);
}

// ---- fixed-order reduction of the per-block double partials + the per-channel BatchNorm algebra.
// One block per channel; replaces ~30 tiny element-wise launches the same math costs in PyTorch.
enum { FIN_STATS = 0, FIN_COEFFS = 1 };

struct FinArgs {
  const double *partial;  // [G, Co, kPartialW]
  int G, Co;
  double count;
  float eps, momentum;
  const float *gamma, *beta, *mean_in, *invstd_in;
  float *running_mean, *running_var;
  long long *num_batches_tracked;  // nn.BatchNorm's step counter, bumped by the block of channel 0 (may be null)
  double *sums;           // [Co,6]: S_a = sum y*rel_a, R_a = sum rel_a  (STATS writes, COEFFS reads)
  float *o0, *o1, *o2, *o3, *o4, *o5;
};

// channel c by the 256 threads t = 0 .. 255 of one thread group; s_red [4][8] and s_tot [8] are the group's scratch.
// `live` false: the group only keeps the barriers company (a channel index past the end in a multi-channel block).
template <int MODE>
__device__ __forceinline__ void finalize_channel(const FinArgs &a, int c, int t, bool live, double (*s_red)[8], double *s_tot) {
  // thread t sums column k = t & 7 of the partial records g = t >> 3 (mod 32): a wave reads 8 whole 64-byte records
  // per load, kFinBatch loads are in flight per thread (the sums are a chain of memory round trips otherwise:
  // measured 16.5 us for 1024 records x 64 channels one at a time, 5.8-6 us eight at a time; 32 at a time -- the metric
  // shape's 1024 records in ONE round trip -- measured the same 6.1-6.5 us: what is left is the launch and 64 blocks
  // pulling 64 KB each in 64-byte pieces), and the 32 partial sums of a column are folded in a fixed order -- lanes by
  // shuffle, then the four waves.
  constexpr int kFinBatch = 8;
  const int k = t & 7, gl = t >> 3;
  double acc = 0.0;
  for (int g0 = 0; g0 < a.G; g0 += 32 * kFinBatch) {
    double v[kFinBatch];
#pragma unroll
    for (int u = 0; u < kFinBatch; ++u) {
      const int g = g0 + u * 32 + gl;
      v[u] = a.partial[((size_t)(g < a.G ? g : a.G - 1) * a.Co + c) * kPartialW + k];
    }
#pragma unroll
    for (int u = 0; u < kFinBatch; ++u)
      if (g0 + u * 32 + gl < a.G) acc += v[u];
  }
  acc += __shfl_xor(acc, 8, CL3D_WAVE);
  acc += __shfl_xor(acc, 16, CL3D_WAVE);
  acc += __shfl_xor(acc, 32, CL3D_WAVE);
  if ((t & 63) < 8) s_red[t >> 6][k] = acc;
  __syncthreads();
  if (t < 8) s_tot[k] = ((s_red[0][k] + s_red[1][k]) + s_red[2][k]) + s_red[3][k];
  __syncthreads();
  if (t != 0 || !live) return;
  if constexpr (MODE == FIN_STATS) {
    const double s0 = s_tot[0], s1 = s_tot[1];
    const double mean = s0 / a.count;
    double var = s1 / a.count - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double invstd = 1.0 / sqrt(var + (double)a.eps);
    const double scale = (double)a.gamma[c] * invstd;
    a.o0[c] = (float)scale;
    a.o1[c] = (float)((double)a.beta[c] - mean * scale);
    a.o2[c] = (float)mean;
    a.o3[c] = (float)invstd;
#pragma unroll
    for (int k = 0; k < 6; ++k) a.sums[c * 6 + k] = s_tot[2 + k];
    if (a.running_mean != nullptr) {  // nn.BatchNorm2d: running = (1-m) running + m batch, unbiased variance
      const double unbiased = var * (a.count / (a.count > 1.0 ? a.count - 1.0 : 1.0));
      a.running_mean[c] = a.running_mean[c] * (1.0f - a.momentum) + a.momentum * (float)mean;
      a.running_var[c] = a.running_var[c] * (1.0f - a.momentum) + a.momentum * (float)unbiased;
    }
    if (c == 0 && a.num_batches_tracked != nullptr) *a.num_batches_tracked += 1;
  } else {
    // BatchNorm backward, affine in y:  dy = A dz [k = k*] + Bc + D y   (s0 = sum dz = d beta, s1 = sum dz*xhat = d gamma)
    const double s0 = s_tot[0], s1 = s_tot[1];
    const double invstd = (double)a.invstd_in[c], mean = (double)a.mean_in[c];
    const double A = (double)a.gamma[c] * invstd;
    const double D = -A * invstd * s1 / a.count;
    const double Bc = -A * s0 / a.count - D * mean;
    a.o0[c] = (float)A;
    a.o1[c] = (float)Bc;
    a.o2[c] = (float)D;
    a.o3[c] = (float)s1;  // d gamma
    a.o4[c] = (float)s0;  // d beta
    // d W_r[c][a] = sum_{slots} dy * rel_a = A T_a + Bc R_a + D S_a
#pragma unroll
    for (int k = 0; k < 3; ++k)
      a.o5[c * 3 + k] = (float)(A * s_tot[2 + k] + Bc * a.sums[c * 6 + 3 + k] + D * a.sums[c * 6 + k]);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void pwmlp_finalize_kernel(FinArgs a) {
  __shared__ double s_red[4][8];
  __shared__ double s_tot[8];
  finalize_channel<MODE>(a, blockIdx.x, threadIdx.x, true, s_red, s_tot);
}

// The arg-max scatter and the BatchNorm-backward algebra in ONE launch: both only need what bwd_rows left, and as two
// launches of the captured step they are 10.6 + 6.5 us one after the other (a fork costs more than either, DESIGN 3.2
// "Round 5").  Slab z = 0 of the grid holds the algebra -- a 1024-thread block = four 256-thread groups = four channels,
// dispatched first -- slabs z = 1 .. B the scatter blocks of cloud z - 1.  The scatter fills every CU with one block
// (128 KB of LDS), so the Co / 4 algebra blocks delay as many scatter blocks by their ~4 us instead of the whole step by
// a launch.
__global__ __launch_bounds__(1024) void pwmlp_hit_coeffs_kernel(HitArgs h, FinArgs f) {
  extern __shared__ double hacc[];  // [4][T]
  if (blockIdx.z == 0) {
    if (blockIdx.y != 0) return;
    __shared__ double s_red[4][4][8];
    __shared__ double s_tot[4][8];
    const int g = threadIdx.x >> 8;
    const int c = blockIdx.x * 4 + g;
    finalize_channel<FIN_COEFFS>(f, c < f.Co ? c : f.Co - 1, threadIdx.x & 255, c < f.Co, s_red[g], s_tot[g]);
    return;
  }
  hit_block(h, blockIdx.z - 1, hacc);
}

// {rel, centre index} per slot for callers of cl3d_pwmlp_fwd that ask for the records (element-wise; the engine's own
// backward passes rebuild rel from the coordinates and do not use it)
__global__ __launch_bounds__(256) void pwmlp_slotrec_kernel(const float *__restrict__ query_xyz,
                                                            const float *__restrict__ support_xyz,
                                                            const int *__restrict__ idx, int B, int N, int M, int K,
                                                            float inv_radius, float4 *__restrict__ slotrec) {
  const long long total = (long long)B * M * K;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long bj = e / K;
    const int b = (int)(bj / M);
    const int i = idx[e];
    const float *sp = support_xyz + ((size_t)b * N + i) * 3;
    const float *qp = query_xyz + (size_t)bj * 3;
    slotrec[e] = make_float4((sp[0] - qp[0]) * inv_radius, (sp[1] - qp[1]) * inv_radius, (sp[2] - qp[2]) * inv_radius,
                             __int_as_float(idx[bj * K]));
  }
}

static int rows_chunks(int Co) { return (Co >> 6) + __builtin_popcount(Co & 63); }

static int pw_check(const PwArgs &a, const char *who) {
  if (a.B < 0 || a.N < 1 || a.M < 1 || a.K < 1 || a.Co < 1) return fail(CL3D_E_INVALID, "%s: bad sizes", who);
  if (a.K > 255) return fail(CL3D_E_UNSUPPORTED, "%s: nsample=%d > 255 (arg-max is stored in a byte)", who, a.K);
  if ((long long)a.M * a.K > 0x7fffffffLL) return fail(CL3D_E_UNSUPPORTED, "%s: M*K too large", who);
  if ((long long)a.N * a.Co * 8 > 0xffffffffLL)  // row offsets inside a cloud's [N, 2Co] table are 32-bit byte offsets
    return fail(CL3D_E_UNSUPPORTED, "%s: N*Co too large", who);
  return CL3D_OK;
}

// queries a lane group walks per tile (template parameter QPG of the query kernels).  Measured at the metric shape
// (round 3): alone, 2 is the faster kernel (68.9 vs 71.1 us: half the barriers and staging round trips per slot);
// inside the replayed step, next to the CSR build on the other queue, 1 is (0.3623-0.3635 vs 0.3791-0.3799 ms per
// step on two boxes).  1 is the default; CL3D_PW_QPG=2 selects the other for A/B timing.
static LaneMap pw_lane_map(int Co, int K, int V, int nacc, int qpg, size_t *lds_out) {
  // (<= 48 lanes per row: the waves' double accumulators -- 4 x L x V x nacc x 8 B -- have to fit beside the slot tile)
  LaneMap m = pick_lane_map(Co, V, 48);
  if (m.QW > 16) {
    m.QW = 16;
    m.L = 4;
    m.chunks = ((Co + V - 1) / V + m.L - 1) / m.L;
  }
  for (;;) {
    const size_t tq = 4 * (size_t)m.QW * qpg;
    const size_t lds = tq * (K + 2) * sizeof(float4) + (size_t)4 * m.L * V * nacc * sizeof(double);
    if (lds <= 60 * 1024 || m.QW == 1) {
      *lds_out = lds;
      return m;
    }
    m.QW -= 1;
  }
}

bool pwmlp_supported(int K, int Co) {
  if (K < 1 || K > 255 || Co < 1) return false;
  const int V = (Co % 4 == 0) ? 4 : 1;
  size_t lds = 0;
  pw_lane_map(Co, K, V, 8, 1, &lds);  // the training pass carries the most accumulators
  return lds <= 64 * 1024;
}

template <int MODE>
static int launch_query(PwArgs &a, int nacc, int n_partials, hipStream_t st, const char *who) {
  const int V = (a.Co % 4 == 0) ? 4 : 1;
  size_t lds = 0;
  const LaneMap m = pw_lane_map(a.Co, a.K, V, nacc, 1, &lds);
  if (lds > 64 * 1024) return fail(CL3D_E_UNSUPPORTED, "%s: nsample=%d needs %zu B of LDS", who, a.K, lds);
  a.L = m.L; a.QW = m.QW; a.chunks = m.chunks;
  a.kmagic = div_magic(a.K);
  const long long tiles = (long long)a.B * ceil_div(a.M, 4 * m.QW);
  const int gx = nacc > 0 ? n_partials : round_grid(tiles, 8192);
  // measured at the metric shape (TRAIN, round 2): 8 gathers in flight at 3 waves/SIMD 87 us, 6 at 3 89 us, 4 at 4 76 us:
  // occupancy buys more than depth per wave (software-pipelining the walk -- batch n+1's gathers issued before batch n
  // is consumed, 2 or 3 waves per SIMD -- measured 95-120 us).  K % 4 == 0: the slot walk in two rotating pairs (round
  // 3: 69.1 -> 64.5 us against the batch walk, which the other shapes keep).  Two queries per lane group (round 3's
  // QPG = 2) measured slower on the replayed step (0.355 against 0.335 ms) and is gone.
  if (V == 4 && (a.K & 3) == 0)
    hipLaunchKernelGGL((pwmlp_query_kernel<MODE, 4, 4, 4, 1, true>), dim3(gx, m.chunks), dim3(256), lds, st, a);
  else if (V == 4)
    hipLaunchKernelGGL((pwmlp_query_kernel<MODE, 4, 4, 4, 1>), dim3(gx, m.chunks), dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL((pwmlp_query_kernel<MODE, 1, 8, 4, 1>), dim3(gx, m.chunks), dim3(256), lds, st, a);
  return check_launch(who);
}

// ---- weight plumbing of the factored contraction (replaces ~8 launch-latency-sized PyTorch kernels per step)
// W [Co, 3+2C] = [W_r | W_c | W_d]  ->  wr [Co,3]  and the per-point GEMM weight  wcat [2Co, C] = [W_d ; W_c - W_d]
__global__ __launch_bounds__(256) void pwmlp_split_weight_kernel(const float *__restrict__ W, int Co, int C,
                                                                 float *__restrict__ wr, float *__restrict__ wcat) {
  const int ld = 3 + 2 * C;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < Co * ld; e += gridDim.x * 256) {
    const int o = e / ld, k = e - o * ld;
    const float v = W[e];
    if (k < 3) {
      wr[o * 3 + k] = v;
    } else if (k < 3 + C) {                       // W_c: lower half of wcat is W_c - W_d
      wcat[(size_t)(Co + o) * C + (k - 3)] = v - W[(size_t)o * ld + k + C];
    } else {                                      // W_d: upper half
      wcat[(size_t)o * C + (k - 3 - C)] = v;
    }
  }
}

// d W [Co, 3+2C] from d wr [Co,3] and the per-cloud products dwb [B, C, 2Co] (= F_b G_b, summed over b in
// order here):  d W_c = bot,  d W_d = top - bot  with  top = d wcat[:Co], bot = d wcat[Co:]
// A block owns a 64(o) x TC(c) tile: rows of dwb are read along o (coalesced; an element-per-thread walk along
// dW's rows reads dwb with a 2Co stride and took 764 us on the widest layer, C = Co = 1152, B = 16), summed over
// b in order, turned through LDS and written along c.  TC trades write segment length (4 TC bytes) for the number
// of blocks; the reads are B times the writes, so blocks win: measured B = 16, C = Co = 72/144/288/576/1152:
// TC=4  4.1 / 4.0 / 4.5 / 10.0 / 33.6 us,  TC=16  9.9 / 9.9 / 10.2 / 22.8 / 29.0 us (5.9 TB/s).
constexpr int kMergeO = 64;
template <int TC>
__global__ __launch_bounds__(256) void pwmlp_merge_weight_grad_tiled_kernel(const float *__restrict__ dwr,
                                                                            const float *__restrict__ dwb, int B, int Co,
                                                                            int C, float *__restrict__ dW) {
  __shared__ float s_top[TC][kMergeO + 1];
  __shared__ float s_bot[TC][kMergeO + 1];
  const int ld = 3 + 2 * C;
  const int o0 = blockIdx.x * kMergeO, c0 = blockIdx.y * TC;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // ty in 0..3
  constexpr int R = TC / 4;                                        // c rows per thread
  float top[R], bot[R];
#pragma unroll
  for (int i = 0; i < R; ++i) top[i] = bot[i] = 0.f;
  const int o = o0 + tx;
  if (o < Co) {
    for (int b = 0; b < B; ++b) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int c = c0 + ty + 4 * i;
        if (c < C) {
          const float *row = dwb + ((size_t)b * C + c) * 2 * Co;
          top[i] += row[o];
          bot[i] += row[Co + o];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    s_top[ty + 4 * i][tx] = top[i];
    s_bot[ty + 4 * i][tx] = bot[i];
  }
  __syncthreads();
  const int cl = threadIdx.x % TC, c = c0 + cl;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int ol = threadIdx.x / TC + (256 / TC) * j, oo = o0 + ol;
    if (oo < Co && c < C) {
      const float t = s_top[cl][ol], bt = s_bot[cl][ol];
      dW[(size_t)oo * ld + 3 + c] = bt;
      dW[(size_t)oo * ld + 3 + C + c] = t - bt;
    }
  }
  if (blockIdx.y == 0 && threadIdx.x < 3 * kMergeO) {
    const int oo = o0 + threadIdx.x / 3, k = threadIdx.x % 3;
    if (oo < Co) dW[(size_t)oo * ld + k] = dwr ? dwr[oo * 3 + k] : 0.f;
  }
}

}  // namespace cl3d

extern "C" int cl3d_pwmlp_split_weight(const float *W, int Co, int C, float *wr, float *wcat, cl3d_stream_t stream) {
  CL3D_REQUIRE(W && wr && wcat && Co >= 1 && C >= 1, "pwmlp_split_weight: bad arguments");
  const int n = Co * (3 + 2 * C);
  hipLaunchKernelGGL(cl3d::pwmlp_split_weight_kernel, dim3(cl3d::ceil_div(n, 256) < 1024 ? cl3d::ceil_div(n, 256) : 1024),
                     dim3(256), 0, (hipStream_t)stream, W, Co, C, wr, wcat);
  return cl3d::check_launch("cl3d_pwmlp_split_weight");
}

extern "C" int cl3d_pwmlp_merge_weight_grad(const float *dwr, const float *dwb, int B, int Co, int C, float *dW,
                                            cl3d_stream_t stream) {
  CL3D_REQUIRE(dwb && dW && B >= 1 && Co >= 1 && C >= 1, "pwmlp_merge_weight_grad: bad arguments");
  const int gx = cl3d::ceil_div(Co, cl3d::kMergeO);
  const hipStream_t st = (hipStream_t)stream;
  if ((long long)gx * cl3d::ceil_div(C, 16) >= 1024)
    hipLaunchKernelGGL(cl3d::pwmlp_merge_weight_grad_tiled_kernel<16>, dim3(gx, cl3d::ceil_div(C, 16)), dim3(256), 0,
                       st, dwr, dwb, B, Co, C, dW);
  else
    hipLaunchKernelGGL(cl3d::pwmlp_merge_weight_grad_tiled_kernel<4>, dim3(gx, cl3d::ceil_div(C, 4)), dim3(256), 0, st,
                       dwr, dwb, B, Co, C, dW);
  return cl3d::check_launch("cl3d_pwmlp_merge_weight_grad");
}

// The gather passes run gx x chunks workgroups (chunks = passes over the channel axis, on gridDim.y), four resident per CU.
// Where a workgroup would get one or two tiles, gx x chunks is kept at or below the 1024 that are resident at once: ONE
// generation of persistent workgroups with more tiles each instead of two generations of one or two (round 6, config 2's
// 144-channel layers at 16 384 points: backbone step bf16 5.61 / 5.70 / 5.62 against 5.70 / 5.69 / 5.67 ms, f32 neutral;
// profiles/r06/session27_summary.txt).  With many tiles per workgroup the two generations cost nothing and the
// chunk-by-chunk order keeps an XCD's L2 on one half of the rows (144 channels x 65 536 points: 0.829 against 0.849 ms).
static int pw_resident_cap(int chunks, long long tiles) {
  if (chunks <= 1 || tiles >= 2048) return 1024;
  const int cap = 1024 / chunks;
  return cap < 8 ? 8 : cap & ~7;
}

extern "C" int cl3d_pwmlp_partials(int B, int M, int Co) {
  size_t lds = 0;
  const int chunks = Co >= 1 ? cl3d::pw_lane_map(Co, 32, (Co % 4 == 0) ? 4 : 1, 8, 1, &lds).chunks : 1;  // (chunks depend on Co only)
  const long long want = ((long long)B * M + 7) / 8;  // (at most two 16-query tiles per workgroup)
  return cl3d::round_grid(want, pw_resident_cap(chunks, want / 2));
}

extern "C" int cl3d_pwmlp_stats(const float *query_xyz, const float *support_xyz, const int32_t *idx,
                                const float *ght, const float *wr, const float *gamma, int B, int N, int M,
                                int K, int Co, float radius, float *ystar_t, unsigned char *kstar_t, float *sy_t,
                                double *partial, int n_partials, cl3d_stream_t stream) {
  using namespace cl3d;
  PwArgs a{};
  a.query_xyz = query_xyz; a.support_xyz = support_xyz; a.idx = idx; a.ght = ght; a.wr = wr; a.v0 = gamma;
  a.ystar_t = ystar_t; a.kstar_out = kstar_t; a.sy_t = sy_t;
  a.partial = partial;
  a.B = B; a.N = N; a.M = M; a.K = K; a.Co = Co; a.inv_radius = 1.0f / radius;
  int rc = pw_check(a, "pwmlp_stats");
  if (rc != CL3D_OK) return rc;
  CL3D_REQUIRE(query_xyz && support_xyz && idx && ght && wr && gamma && ystar_t && kstar_t && sy_t && partial,
               "pwmlp_stats: null pointer");
  CL3D_REQUIRE(n_partials == cl3d_pwmlp_partials(B, M, Co), "pwmlp_stats: partial buffer must have cl3d_pwmlp_partials() blocks");
  if (B == 0) return CL3D_OK;
  return launch_query<PW_TRAIN>(a, 8, n_partials, (hipStream_t)stream, "cl3d_pwmlp_stats");
}

extern "C" int cl3d_pwmlp_finalize_stats(const double *partial, int n_partials, int Co, double count, float eps,
                                         float momentum, const float *gamma, const float *beta,
                                         float *running_mean, float *running_var, int64_t *num_batches_tracked,
                                         float *scale, float *shift, float *mean, float *invstd, double *sums,
                                         cl3d_stream_t stream) {
  CL3D_REQUIRE(partial && gamma && beta && scale && shift && mean && invstd && sums && n_partials > 0 && Co > 0 && count > 0,
               "pwmlp_finalize_stats: bad arguments");
  cl3d::FinArgs a{};
  a.partial = partial; a.G = n_partials; a.Co = Co; a.count = count; a.eps = eps; a.momentum = momentum;
  a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var;
  a.num_batches_tracked = reinterpret_cast<long long *>(num_batches_tracked);
  a.o0 = scale; a.o1 = shift; a.o2 = mean; a.o3 = invstd; a.sums = sums;
  hipLaunchKernelGGL((cl3d::pwmlp_finalize_kernel<cl3d::FIN_STATS>), dim3(Co), dim3(256), 0, (hipStream_t)stream, a);
  return cl3d::check_launch("cl3d_pwmlp_finalize_stats");
}

extern "C" int cl3d_pwmlp_apply(const float *ystar_t, const float *scale, const float *shift, int B, int M,
                                int Co, float *out, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 0 && M >= 1 && Co >= 1, "pwmlp_apply: bad sizes");
  CL3D_REQUIRE(ystar_t && scale && shift && out, "pwmlp_apply: null pointer");
  if (B == 0) return CL3D_OK;
  RowArgs a{};
  a.ystar_t = ystar_t; a.scale = scale; a.shift = shift; a.out = out; a.B = B; a.M = M; a.Co = Co; a.K = 1;
  const int gx = round_grid((long long)B * ceil_div(M, 64), 8192);
  if (rows64_covers(a, false))
    hipLaunchKernelGGL((pwmlp_rows64_kernel<ROWS_APPLY>), dim3(gx, Co / 64), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((pwmlp_rows_kernel<ROWS_APPLY>), dim3(gx, rows_chunks(Co)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("cl3d_pwmlp_apply");
}

extern "C" int cl3d_pwmlp_bn_backward_coeffs(const double *partial, int n_partials, int Co, double count,
                                             const float *gamma, const float *mean, const float *invstd,
                                             const double *sums, float *cA, float *cB, float *cD, float *dgamma,
                                             float *dbeta, float *dwr, cl3d_stream_t stream) {
  CL3D_REQUIRE(partial && gamma && mean && invstd && sums && cA && cB && cD && dgamma && dbeta && dwr && n_partials > 0 &&
                   Co > 0 && count > 0,
               "pwmlp_bn_backward_coeffs: bad arguments");
  cl3d::FinArgs a{};
  a.partial = partial; a.G = n_partials; a.Co = Co; a.count = count; a.gamma = gamma; a.mean_in = mean;
  a.invstd_in = invstd; a.sums = const_cast<double *>(sums);
  a.o0 = cA; a.o1 = cB; a.o2 = cD; a.o3 = dgamma; a.o4 = dbeta; a.o5 = dwr;
  hipLaunchKernelGGL((cl3d::pwmlp_finalize_kernel<cl3d::FIN_COEFFS>), dim3(Co), dim3(256), 0, (hipStream_t)stream, a);
  return cl3d::check_launch("cl3d_pwmlp_bn_backward_coeffs");
}

extern "C" int cl3d_pwmlp_fwd(const float *query_xyz, const float *support_xyz, const int32_t *idx,
                              const float *ght, const float *wr, const float *scale, const float *shift,
                              int B, int N, int M, int K, int Co, float radius, float *out,
                              int out_channel_major, unsigned char *kstar_t, float *slotrec,
                              cl3d_stream_t stream) {
  using namespace cl3d;
  PwArgs a{};
  a.query_xyz = query_xyz; a.support_xyz = support_xyz; a.idx = idx; a.ght = ght; a.wr = wr;
  a.v0 = scale; a.v1 = shift; a.out_t = out; a.out_channel_major = out_channel_major; a.kstar_out = kstar_t;
  a.B = B; a.N = N; a.M = M; a.K = K; a.Co = Co; a.inv_radius = 1.0f / radius;
  int rc = pw_check(a, "pwmlp_fwd");
  if (rc != CL3D_OK) return rc;
  CL3D_REQUIRE(query_xyz && support_xyz && idx && ght && wr && scale && shift && out, "pwmlp_fwd: null pointer");
  if (B == 0) return CL3D_OK;
  if (slotrec != nullptr) {
    const long long total = (long long)B * M * K;
    hipLaunchKernelGGL(pwmlp_slotrec_kernel, dim3(round_grid((total + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                       query_xyz, support_xyz, idx, B, N, M, K, a.inv_radius, reinterpret_cast<float4 *>(slotrec));
    rc = check_launch("cl3d_pwmlp_fwd(slotrec)");
    if (rc != CL3D_OK) return rc;
  }
  return launch_query<PW_FWD>(a, 0, 0, (hipStream_t)stream, "cl3d_pwmlp_fwd");
}

extern "C" int cl3d_pwmlp_bwd_rows(const float *gout, int gout_channel_major, const float *ystar_t,
                                   const unsigned char *kstar_t, const int32_t *idx, const float *query_xyz,
                                   const float *support_xyz, float radius, const float *scale, const float *shift,
                                   const float *mean, const float *invstd, int B, int N, int M, int K, int Co,
                                   float *dz_cm, int32_t *ts_cm, float *dz_t, float *qtab, double *partial,
                                   int n_partials, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 0 && N >= 1 && M >= 1 && K >= 1 && K <= 255 && Co >= 1 && radius > 0.f, "pwmlp_bwd_rows: bad sizes");
  CL3D_REQUIRE(gout && ystar_t && kstar_t && idx && query_xyz && support_xyz && scale && shift && mean && invstd && dz_cm && ts_cm &&
                   dz_t && partial,
               "pwmlp_bwd_rows: null pointer");  // qtab may be null: only cl3d_pwmlp_bwd_support reads it
  CL3D_REQUIRE(n_partials == cl3d_pwmlp_partials(B, M, Co), "pwmlp_bwd_rows: wrong partial block count");
  if (B == 0) return CL3D_OK;
  RowArgs a{};
  a.idx = idx; a.dz_cm = dz_cm; a.ts_cm = ts_cm; a.dz_t = dz_t; a.qtab = reinterpret_cast<float4 *>(qtab);
  a.gout = gout; a.gout_channel_major = gout_channel_major; a.ystar_t = ystar_t; a.kstar_t = kstar_t;
  a.query_xyz = query_xyz; a.support_xyz = support_xyz; a.inv_radius = 1.0f / radius; a.N = N;
  a.scale = scale; a.shift = shift; a.mean = mean;
  a.invstd = invstd; a.partial = partial; a.B = B; a.M = M; a.K = K; a.Co = Co;
  if (rows64_covers(a, true))
    hipLaunchKernelGGL((pwmlp_rows64_kernel<ROWS_BWD>), dim3(n_partials, Co / 64), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((pwmlp_rows_kernel<ROWS_BWD>), dim3(n_partials, rows_chunks(Co)), dim3(256), 0, (hipStream_t)stream,
                       a);
  return check_launch("cl3d_pwmlp_bwd_rows");
}

static int launch_hit_wide(const cl3d::HitArgs &a, int chb, hipStream_t st, const char *who) {
  using namespace cl3d;
  static std::atomic<unsigned long long> granted{0};
  int rc_lds = lds_opt_in(granted, reinterpret_cast<const void *>(pwmlp_hit_wide_kernel), 128 * 1024, who);
  if (rc_lds != CL3D_OK) return rc_lds;
  hipLaunchKernelGGL(pwmlp_hit_wide_kernel, dim3(ceil_div(a.Co, chb), 1, a.B), dim3(1024), (size_t)chb * a.N * sizeof(double), st,
                     a, chb);
  return check_launch(who);
}

extern "C" int cl3d_pwmlp_bwd_hits(const float *dz_cm, const int32_t *ts_cm, int B, int N, int M, int Co,
                                   float *hit_cm, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 0 && N >= 1 && M >= 1 && Co >= 1, "pwmlp_bwd_hits: bad sizes");
  CL3D_REQUIRE(dz_cm && ts_cm && hit_cm, "pwmlp_bwd_hits: null pointer");
  if (B == 0) return CL3D_OK;
  CL3D_REQUIRE(B <= 65535, "pwmlp_bwd_hits: B exceeds grid.z limit");
  static std::atomic<unsigned long long> hit_granted{0};
  int rc_lds = lds_opt_in(hit_granted, reinterpret_cast<const void *>(pwmlp_hit_kernel), 128 * 1024, "pwmlp_bwd_hits");
  if (rc_lds != CL3D_OK) return rc_lds;
  HitArgs a{};
  a.dz_cm = dz_cm; a.ts_cm = ts_cm; a.hit_cm = hit_cm; a.B = B; a.N = N; a.M = M; a.Co = Co;
  if (const int chb = hit_wide_rows(N, M)) return launch_hit_wide(a, chb, (hipStream_t)stream, "cl3d_pwmlp_bwd_hits");
  a.T = N < 4096 ? N : 4096;  // 4 channels x T doubles = 128 KiB
  const int ntiles = ceil_div(N, a.T);
  CL3D_REQUIRE(ntiles <= 65535, "pwmlp_bwd_hits: N too large");
  hipLaunchKernelGGL(pwmlp_hit_kernel, dim3(ceil_div(Co, 4), ntiles, B), dim3(1024), (size_t)4 * a.T * sizeof(double),
                     (hipStream_t)stream, a);
  return check_launch("cl3d_pwmlp_bwd_hits");
}

extern "C" int cl3d_pwmlp_bwd_hits_coeffs(const double *partial, int n_partials, double count, const float *gamma,
                                          const float *mean, const float *invstd, const double *sums, float *cA,
                                          float *cB, float *cD, float *dgamma, float *dbeta, float *dwr,
                                          const float *dz_cm, const int32_t *ts_cm, int B, int N, int M, int Co,
                                          float *hit_cm, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(B >= 1 && N >= 1 && M >= 1 && Co >= 1 && n_partials > 0 && count > 0, "pwmlp_bwd_hits_coeffs: bad sizes");
  CL3D_REQUIRE(partial && gamma && mean && invstd && sums && cA && cB && cD && dgamma && dbeta && dwr && dz_cm && ts_cm && hit_cm,
               "pwmlp_bwd_hits_coeffs: null pointer");
  CL3D_REQUIRE(B <= 65534, "pwmlp_bwd_hits_coeffs: B exceeds grid.z limit");
  static std::atomic<unsigned long long> granted{0};
  int rc_lds = lds_opt_in(granted, reinterpret_cast<const void *>(pwmlp_hit_coeffs_kernel), 128 * 1024, "pwmlp_bwd_hits_coeffs");
  if (rc_lds != CL3D_OK) return rc_lds;
  if (hit_wide_rows(N, M)) {
    // few queries per cloud: the scatter is a handful of wide blocks, and the one-launch form would carry Co / 4 x B
    // blocks for it -- two small launches instead
    const int rc = cl3d_pwmlp_bn_backward_coeffs(partial, n_partials, Co, count, gamma, mean, invstd, sums, cA, cB, cD, dgamma,
                                                 dbeta, dwr, stream);
    return rc != CL3D_OK ? rc : cl3d_pwmlp_bwd_hits(dz_cm, ts_cm, B, N, M, Co, hit_cm, stream);
  }
  HitArgs h{};
  h.dz_cm = dz_cm; h.ts_cm = ts_cm; h.hit_cm = hit_cm; h.B = B; h.N = N; h.M = M; h.Co = Co;
  h.T = N < 4096 ? N : 4096;  // 4 channels x T doubles = 128 KiB
  const int ntiles = ceil_div(N, h.T);
  CL3D_REQUIRE(ntiles <= 65535, "pwmlp_bwd_hits_coeffs: N too large");
  FinArgs f{};
  f.partial = partial; f.G = n_partials; f.Co = Co; f.count = count; f.gamma = gamma; f.mean_in = mean;
  f.invstd_in = invstd; f.sums = const_cast<double *>(sums);
  f.o0 = cA; f.o1 = cB; f.o2 = cD; f.o3 = dgamma; f.o4 = dbeta; f.o5 = dwr;
  hipLaunchKernelGGL(pwmlp_hit_coeffs_kernel, dim3(ceil_div(Co, 4), ntiles, B + 1), dim3(1024),
                     (size_t)4 * h.T * sizeof(double), (hipStream_t)stream, h, f);
  return check_launch("cl3d_pwmlp_bwd_hits_coeffs");
}

extern "C" int cl3d_pwmlp_bwd_support(const float *ght, const float *wr, const float *cA, const float *cB,
                                      const float *cD, const float *hit_cm, const float *dz_t, const float *sy_t,
                                      const float *qtab, const float *support_xyz, float radius,
                                      const int32_t *inv_off, const int32_t *inv_slots, int B, int N, int M, int K,
                                      int Co, float *dght, cl3d_stream_t stream) {
  using namespace cl3d;
  PwArgs a{};
  a.ght = ght; a.wr = wr; a.v0 = cA; a.v1 = cB; a.v2 = cD; a.hit_cm = hit_cm; a.dz_t = dz_t; a.sy_in = sy_t;
  a.qtab = reinterpret_cast<const float4 *>(qtab); a.support_xyz = support_xyz; a.inv_radius = 1.0f / radius;
  a.inv_off = inv_off; a.inv_slots = inv_slots; a.dght = dght;
  a.B = B; a.N = N; a.M = M; a.K = K; a.Co = Co;
  a.kmagic = div_magic(K, (long long)M * K);
  int rc = pw_check(a, "pwmlp_bwd_support");
  if (rc != CL3D_OK) return rc;
  CL3D_REQUIRE(ght && wr && cA && cB && cD && hit_cm && dz_t && sy_t && qtab && support_xyz && inv_off && inv_slots && dght && radius > 0.f,
               "pwmlp_bwd_support: null pointer");
  if ((long long)M * Co * 4 > 0xffffffffLL) return fail(CL3D_E_UNSUPPORTED, "pwmlp_bwd_support: M*Co too large");
  // an entry's row offset carries the "centred" flag in bit 31 (kCentreFlag): offsets inside a cloud's table stay below 2 GiB
  if ((long long)N * Co * 8 > 0x7fffffffLL)
    return fail(CL3D_E_UNSUPPORTED, "pwmlp_bwd_support: N*Co too large for 31-bit row offsets");
  if (B == 0) return CL3D_OK;
  const int V = (Co % 4 == 0) ? 4 : 1;
  // (round 5: a map chosen for the per-ENTRY chain instead of for busy channel lanes -- e.g. 32 lanes per point for Co = 72,
  //  one round per 32-entry list and one channel chunk instead of 9 lanes, four rounds, two chunks -- measured slower on the
  //  config-2 backbone: 59.5 against 56.6 us per launch on average)
  const LaneMap m = pick_lane_map(Co, V);
  a.L = m.L; a.QW = m.QW; a.chunks = m.chunks;
  const long long tiles = (long long)B * ceil_div(N, 4 * m.QW);
  if (tiles > 0x7fffffffLL) return fail(CL3D_E_UNSUPPORTED, "pwmlp_bwd_support: too many tiles");
  if (m.L * V > 256) return fail(CL3D_E_UNSUPPORTED, "pwmlp_bwd_support: %d channels per chunk", m.L * V);
  // persistent: as many workgroups per CU as fit -- counting the channel chunks on gridDim.y where tiles are few -- each
  // pipelines over its tiles
  const int gx = round_grid(tiles, pw_resident_cap(m.chunks, tiles) * CL3D_SUP_WAVES / 4);
  if (V == 4) hipLaunchKernelGGL((pwmlp_support_kernel<4, CL3D_SUP_SB>), dim3(gx, m.chunks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((pwmlp_support_kernel<1, 8>), dim3(gx, m.chunks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("cl3d_pwmlp_bwd_support");
}

// pass.hip -- one C-ABI call per PASS of a PointWiseMLP LocalAggregation in training mode (round 4).
//
// The reference's eager training loop makes one `_ext` call per autograd node (pt_utils.py:16-61) and leaves every
// other operation to PyTorch; the engine's fused operator is a dozen kernels per direction, and launched one C-ABI
// call at a time from Python the host sets the pace (0.41 ms per step at the metric shape against 0.32 ms for the
// same kernels replayed as a HIP graph) -- with the geometry work forked onto side streams from Python it is slower
// still (0.59 ms: stream guards, event objects and allocator bookkeeping cost more than the overlap returns).
// cl3d_pwmlp_train_forward / _backward enqueue EVERY kernel of a pass from here, forks included:
//
//   forward    caller's stream: ball query ... joins the product: statistics pass, BatchNorm algebra, activation +
//              transposition; joins the CSR           |   side 0: weights + per-point product
//                                                     |   side 1: (behind the query) CSR inverse
//   backward   caller's stream: rows pass, BatchNorm backward algebra, arg-max scatter, support-major pass, weight
//              gradient of the per-point product      |   side 0: its data gradient (joined)
// (the longest chain never leaves the caller's stream: a forked piece starts one cross-queue hand-over late)
//
// The side streams and events belong to the library (one set per device, created on first use, non-blocking); a
// stream waits for an event, never the host.  Every buffer -- outputs, intermediates kept for the backward pass,
// scratch -- is the caller's (cl3d_pwmlp_pass): nothing is allocated here, and every fork is joined before the call
// that made it returns (joined = the caller's stream waits; the host never does).
// Inside a stream capture the same calls record the same forks as branches of the caller's graph; outside one, a pass
// called twice in a row with a bit-identical argument block is captured into a launch graph of its own and replayed
// from then on (run_pass below).
#include <atomic>
#include <cstring>
#include <mutex>

#include "cl3d_common.h"

namespace cl3d {

struct PassRuntime {
  hipStream_t side[2] = {nullptr, nullptr};
  hipStream_t cap = nullptr;  // launch graphs are captured here, never on the caller's stream (which may be the legacy one)
  std::mutex use;             // one pass at a time per device: the side streams and events are shared by the calls
  int dev = 0;
  hipEvent_t ev_in = nullptr, ev_bq = nullptr, ev_csr = nullptr, ev_fork = nullptr, ev_w = nullptr;
  bool ok = false;
};

static PassRuntime *pass_runtime() {
  static PassRuntime rt[64];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  PassRuntime &r = rt[dev];
  if (!r.ok) {
    bool good = true;
    for (int k = 0; k < 2; ++k) good = good && hipStreamCreateWithFlags(&r.side[k], hipStreamNonBlocking) == hipSuccess;
    good = good && hipStreamCreateWithFlags(&r.cap, hipStreamNonBlocking) == hipSuccess;
    hipEvent_t *evs[5] = {&r.ev_in, &r.ev_bq, &r.ev_csr, &r.ev_fork, &r.ev_w};
    for (hipEvent_t *e : evs) good = good && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    if (!good) return nullptr;
    r.dev = dev;
    r.ok = true;
  }
  return &r;
}

static int hip_ok(hipError_t e, const char *what) {
  return e == hipSuccess ? CL3D_OK : fail(CL3D_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

// ---- launch graphs.  An eager training loop calls a pass with the SAME argument block step after step (the caller's
// caching allocator hands the same addresses to the same requests), and the host -- a dozen kernel launches and half a
// dozen event calls per pass -- is what paces it.  The second time a pass is called with a bit-identical block (within
// the last kSeen calls of its direction) its launches are captured, on the library's own stream, into a HIP graph, and
// every later call with that block is ONE hipGraphLaunch on the caller's stream.  A kernel's behaviour depends on
// nothing but its arguments, all of which are in the block (sizes, scalars, every pointer), so a replay IS the call it
// was captured from; a block seen once (varying batch shapes, a caller that never repeats addresses) is enqueued directly
// as before.  kGraphSlots graphs per direction and device; a graph is only evicted when it has not been launched for
// kIdleCalls calls (a loop over more distinct blocks than slots falls back to direct launches instead of re-capturing
// every call), and an evicted graph is destroyed kIdleCalls calls later still (its last launch has long drained).
constexpr int kGraphSlots = 8, kSeen = 32, kIdleCalls = 64;
template <class Block>
struct PassGraphs {
  Block key[kGraphSlots];
  hipGraphExec_t exec[kGraphSlots] = {};
  unsigned long long used[kGraphSlots] = {};
  unsigned long long seen[kSeen] = {};  // hashes of the last blocks that were launched directly
  int seen_at = 0;
  hipGraphExec_t retired[kGraphSlots] = {};
  unsigned long long retired_at[kGraphSlots] = {};
  unsigned long long tick = 0;
};
template <class Block>
static PassGraphs<Block> &graphs_of(int dir, int dev) {
  static PassGraphs<Block> g[2][64];
  return g[dir][dev];
}
static std::atomic<int> g_graphs_on{1};
static std::atomic<long long> g_captures{0}, g_replays{0};

static_assert(sizeof(cl3d_pwmlp_pass) == 14 * 4 + 13 * 8 + 4 * sizeof(size_t) + 29 * 8, "cl3d_pwmlp_pass has padding: compare field-wise");

// the block as the launch-graph tables key it: a byte copy with the reserved word (the only bytes no field of the ABI
// defines) cleared, so two calls with the same arguments compare equal whatever the caller left there
static_assert(sizeof(cl3d_reduce_pass) == 16 * 4 + 13 * 8 + 2 * sizeof(size_t) + 10 * 8 + 10 * 8 + 4 * 4,
              "cl3d_reduce_pass has padding: compare field-wise");

static void clear_reserved(cl3d_pwmlp_pass &k) { k.reserved = 0; }
static void clear_reserved(cl3d_reduce_pass &k) { k.reserved = 0; k.reserved2 = 0; }

template <class Block>
static Block block_key(const Block *p) {
  Block k;
  memcpy(&k, p, sizeof(k));
  clear_reserved(k);
  return k;
}

template <class Block>
static unsigned long long block_hash(const Block *p) {  // FNV-1a over the block's bytes (never 0)
  const unsigned char *b = reinterpret_cast<const unsigned char *>(p);
  unsigned long long h = 1469598103934665603ull;
  for (size_t k = 0; k < sizeof(*p); ++k) h = (h ^ b[k]) * 1099511628211ull;
  return h != 0 ? h : 1;
}

// (called with the device's PassRuntime::use lock held: one pass at a time per device uses the side streams and events)
template <class Block, class Enqueue>
static int run_pass(int dir, const Block *p, hipStream_t st, PassRuntime *rt, int dev, Enqueue &&enqueue) {
  if (!g_graphs_on.load()) return enqueue(st);
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return enqueue(st);  // inside the caller's own capture: the launches become nodes of ITS graph
  }
  PassGraphs<Block> &g = graphs_of<Block>(dir, dev);
  const Block key = block_key(p);
  ++g.tick;
  for (int k = 0; k < kGraphSlots; ++k)
    if (g.retired[k] != nullptr && g.retired_at[k] + kIdleCalls < g.tick) {
      (void)hipGraphExecDestroy(g.retired[k]);
      g.retired[k] = nullptr;
    }
  for (int k = 0; k < kGraphSlots; ++k)
    if (g.exec[k] != nullptr && memcmp(&g.key[k], &key, sizeof(key)) == 0) {
      g.used[k] = g.tick;
      g_replays.fetch_add(1);
      return hip_ok(hipGraphLaunch(g.exec[k], st), "pwmlp pass: graph launch");
    }
  const unsigned long long h = block_hash(&key);
  bool seen = false;
  for (int k = 0; k < kSeen; ++k) seen = seen || g.seen[k] == h;
  int slot = -1;  // an empty slot, else one idle for kIdleCalls calls whose predecessor in the retired list is gone
  for (int k = 0; k < kGraphSlots && slot < 0; ++k)
    if (g.exec[k] == nullptr) slot = k;
  if (slot < 0)
    for (int k = 0; k < kGraphSlots; ++k)
      if (g.used[k] + kIdleCalls < g.tick && g.retired[k] == nullptr && (slot < 0 || g.used[k] < g.used[slot])) slot = k;
  if (!seen || slot < 0) {  // first sighting (or nowhere to keep a graph): launch directly, remember the block
    if (!seen) {
      g.seen[g.seen_at] = h;
      g.seen_at = (g.seen_at + 1) % kSeen;
    }
    return enqueue(st);
  }
  // seen before: capture (nothing runs), instantiate, launch
  int rc = hip_ok(hipStreamBeginCapture(rt->cap, hipStreamCaptureModeRelaxed), "pwmlp pass: begin capture");
  if (rc != CL3D_OK) return rc;
  rc = enqueue(rt->cap);
  hipGraph_t graph = nullptr;
  const hipError_t ee = hipStreamEndCapture(rt->cap, &graph);
  if (rc == CL3D_OK) rc = hip_ok(ee, "pwmlp pass: end capture");
  hipGraphExec_t exec = nullptr;
  if (rc == CL3D_OK) rc = hip_ok(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0), "pwmlp pass: instantiate");
  if (graph != nullptr) (void)hipGraphDestroy(graph);
  if (rc != CL3D_OK) {
    (void)hipGetLastError();
    return rc;
  }
  if (g.exec[slot] != nullptr) {  // (idle for kIdleCalls calls; destroyed another kIdleCalls calls from now)
    g.retired[slot] = g.exec[slot];
    g.retired_at[slot] = g.tick;
  }
  memcpy(&g.key[slot], &key, sizeof(key));
  g.exec[slot] = exec;
  g.used[slot] = g.tick;
  g_captures.fetch_add(1);
  return hip_ok(hipGraphLaunch(exec, st), "pwmlp pass: graph launch");
}

}  // namespace cl3d

#define CL3D_TRY(call)             \
  do {                             \
    const int rc_ = (call);        \
    if (rc_ != CL3D_OK) return rc_; \
  } while (0)

static int enqueue_forward(const cl3d_pwmlp_pass *p, hipStream_t st, cl3d::PassRuntime *rt);
static int enqueue_backward(const cl3d_pwmlp_pass *p, hipStream_t st, cl3d::PassRuntime *rt);

extern "C" int cl3d_pwmlp_pass_graphs(int enable) { return cl3d::g_graphs_on.exchange(enable != 0 ? 1 : 0); }

extern "C" int cl3d_pwmlp_pass_graph_stats(long long *captures, long long *replays) {
  if (captures != nullptr) *captures = cl3d::g_captures.load();
  if (replays != nullptr) *replays = cl3d::g_replays.load();
  return CL3D_OK;
}

extern "C" int cl3d_pwmlp_train_forward(const cl3d_pwmlp_pass *p, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(p != nullptr, "pwmlp_train_forward: null argument block");
  CL3D_REQUIRE(p->B >= 0 && p->N >= 1 && p->M >= 1 && p->K >= 1 && p->C >= 1 && p->Co >= 1 && p->radius > 0.f,
               "pwmlp_train_forward: bad sizes");
  CL3D_REQUIRE(p->query_xyz && p->support_xyz && p->query_mask && p->support_mask && p->idx && p->idx_mask && p->features &&
                   p->W && p->gamma && p->beta && p->ght && p->wr && p->wcat && p->ystar && p->sy && p->kstar && p->partial &&
                   p->vec && p->sums && p->out,
               "pwmlp_train_forward: null pointer");
  if (p->B == 0) return CL3D_OK;
  PassRuntime *rt = pass_runtime();
  if (rt == nullptr) return fail(CL3D_E_LAUNCH, "pwmlp_train_forward: side streams could not be created");
  std::lock_guard<std::mutex> use(rt->use);
  return run_pass(0, p, (hipStream_t)stream, rt, rt->dev, [&](hipStream_t s) { return enqueue_forward(p, s, rt); });
}

static int enqueue_forward(const cl3d_pwmlp_pass *p, hipStream_t st, cl3d::PassRuntime *rt) {
  using namespace cl3d;
  const bool want_csr = p->inv_off != nullptr && !p->csr_ready;
  // ---- the LONGEST chain stays on the caller's stream: ball query -> statistics pass -> BatchNorm -> activation.  A piece
  // handed to a side stream starts one cross-queue hand-over (~10 us) late and its join is free once it has finished: so
  // the per-point product (26 us, needs nothing of the geometry) is what forks, and it is done long before the query
  // (55-62 us) is; the CSR build forks behind the query -- BEFORE the statistics pass is enqueued, see below -- and is joined
  // at the very end
  // (the query is ENQUEUED first, the product forked from the point before it: in a captured pass the first-recorded root
  // and its first-recorded dependents get the launch queue)
  CL3D_TRY(hip_ok(hipEventRecord(rt->ev_in, st), "pwmlp_train_forward: event"));
  if (!p->idx_ready)
    CL3D_TRY(cl3d_masked_ordered_ball_query(p->query_xyz, p->support_xyz, p->query_mask, p->support_mask, p->B, p->M, p->N,
                                            p->radius, p->K, p->idx, p->idx_mask, p->bq_ws, p->bq_ws_bytes, st));
  CL3D_TRY(hip_ok(hipStreamWaitEvent(rt->side[0], rt->ev_in, 0), "pwmlp_train_forward: fork"));
  CL3D_TRY(cl3d_pwmlp_point_gemm_fwd(p->features, p->W, p->B, p->C, p->N, p->Co, p->precision, p->ght, p->wr, p->wcat,
                                     p->gemm_ws, p->gemm_ws_bytes, rt->side[0]));
  CL3D_TRY(hip_ok(hipEventRecord(rt->ev_fork, rt->side[0]), "pwmlp_train_forward: event"));
  // The CSR build is enqueued BEFORE the statistics pass (round 5, second session; fused.pointwise_mlp has the same order
  // and the measurements): a captured pass is laid out depth-first along each node's first-recorded dependent, so the
  // build inherits the query's queue and its count pass runs the moment the query ends -- 11 us alone instead of 60 us
  // squeezed in behind the statistics pass's workgroups -- while the statistics pass joins the product's queue (one
  // cross-queue hand-over either way: behind the product before, behind the query now).
  if (want_csr) {
    CL3D_REQUIRE(p->inv_slots != nullptr, "pwmlp_train_forward: null inv_slots");
    CL3D_TRY(hip_ok(hipEventRecord(rt->ev_bq, st), "pwmlp_train_forward: event"));
    CL3D_TRY(hip_ok(hipStreamWaitEvent(rt->side[1], rt->ev_bq, 0), "pwmlp_train_forward: fork"));
    CL3D_TRY(cl3d_build_inverse_index(p->idx, p->B, p->N, p->M * p->K, p->inv_off, p->inv_slots, p->csr_ws, p->csr_ws_bytes,
                                      rt->side[1]));
    CL3D_TRY(hip_ok(hipEventRecord(rt->ev_csr, rt->side[1]), "pwmlp_train_forward: event"));
  }
  CL3D_TRY(hip_ok(hipStreamWaitEvent(st, rt->ev_fork, 0), "pwmlp_train_forward: join"));
  CL3D_TRY(cl3d_pwmlp_stats(p->query_xyz, p->support_xyz, p->idx, p->ght, p->wr, p->gamma, p->B, p->N, p->M, p->K, p->Co,
                            p->radius, p->ystar, p->kstar, p->sy, p->partial, p->n_partials, st));
  float *scale = p->vec, *shift = p->vec + p->Co, *mean = p->vec + 2 * p->Co, *invstd = p->vec + 3 * p->Co;
  CL3D_TRY(cl3d_pwmlp_finalize_stats(p->partial, p->n_partials, p->Co, (double)p->B * p->M * p->K, p->eps, p->momentum,
                                     p->gamma, p->beta, p->running_mean, p->running_var, p->num_batches_tracked, scale, shift,
                                     mean, invstd, p->sums, st));
  CL3D_TRY(cl3d_pwmlp_apply(p->ystar, scale, shift, p->B, p->M, p->Co, p->out, st));
  // the CSR build is joined HERE, behind ~100 us of operator kernels it finished beside: nothing of this pass is still in
  // flight on a side stream once the caller's stream has passed this point, so the caller may drop the buffers of a
  // forward pass whose backward never runs
  if (want_csr) CL3D_TRY(hip_ok(hipStreamWaitEvent(st, rt->ev_csr, 0), "pwmlp_train_forward: join"));
  return CL3D_OK;
}

extern "C" int cl3d_pwmlp_train_backward(const cl3d_pwmlp_pass *p, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(p != nullptr, "pwmlp_train_backward: null argument block");
  CL3D_REQUIRE(p->gout && p->dz_cm && p->ts_cm && p->dz_t && p->qtab && p->partial_b && p->hit && p->coef && p->dwr &&
                   p->dght && p->inv_off && p->inv_slots && p->ystar && p->kstar && p->sy && p->vec && p->sums && p->ght &&
                   p->wr && p->wcat && p->features,
               "pwmlp_train_backward: null pointer");
  if (p->B == 0) return CL3D_OK;
  PassRuntime *rt = pass_runtime();
  if (rt == nullptr) return fail(CL3D_E_LAUNCH, "pwmlp_train_backward: side streams could not be created");
  std::lock_guard<std::mutex> use(rt->use);
  return run_pass(1, p, (hipStream_t)stream, rt, rt->dev, [&](hipStream_t s) { return enqueue_backward(p, s, rt); });
}

static int enqueue_backward(const cl3d_pwmlp_pass *p, hipStream_t st, cl3d::PassRuntime *rt) {
  using namespace cl3d;
  const int Co = p->Co;
  const float *scale = p->vec, *shift = p->vec + Co, *mean = p->vec + 2 * Co, *invstd = p->vec + 3 * Co;
  float *cA = p->coef, *cB = p->coef + Co, *cD = p->coef + 2 * Co, *dgamma = p->coef + 3 * Co, *dbeta = p->coef + 4 * Co;
  CL3D_TRY(cl3d_pwmlp_bwd_rows(p->gout, 1, p->ystar, p->kstar, p->idx, p->query_xyz, p->support_xyz, p->radius, scale, shift,
                               mean, invstd, p->B, p->N, p->M, p->K, Co, p->dz_cm, p->ts_cm, p->dz_t, p->qtab, p->partial_b,
                               p->n_partials, st));
  CL3D_TRY(cl3d_pwmlp_bwd_hits_coeffs(p->partial_b, p->n_partials, (double)p->B * p->M * p->K, p->gamma, mean, invstd, p->sums,
                                      cA, cB, cD, dgamma, dbeta, p->dwr, p->dz_cm, p->ts_cm, p->B, p->N, p->M, Co, p->hit, st));
  CL3D_TRY(cl3d_pwmlp_bwd_support(p->ght, p->wr, cA, cB, cD, p->hit, p->dz_t, p->sy, p->qtab, p->support_xyz, p->radius,
                                  p->inv_off, p->inv_slots, p->B, p->N, p->M, p->K, Co, p->dght, st));
  // ---- the two gradient products: ONE kernel over d ght where it covers the shape; otherwise side by side, the longer one (weights: product + slice reduce, ~45 us) on the caller's
  // stream, the data gradient (~25 us) on the side stream -- hand-over included it ends first, the join is free
  if (p->dW != nullptr && p->dfeat != nullptr && cl3d_pwmlp_point_gemm_bwd_fused(p->B, p->C, p->N, Co, p->precision))
    return cl3d_pwmlp_point_gemm_bwd(p->features, nullptr, nullptr, p->dght, p->wcat, p->dwr, p->B, p->C, p->N, Co,
                                     p->precision, p->dfeat, p->dW, p->gemm_ws_w, p->gemm_ws_bytes_b, st);  // one kernel forms both
  const bool fork = p->dW != nullptr && p->dfeat != nullptr;
  hipStream_t dst = st;
  if (fork) {
    CL3D_TRY(hip_ok(hipEventRecord(rt->ev_fork, st), "pwmlp_train_backward: event"));
    CL3D_TRY(hip_ok(hipStreamWaitEvent(rt->side[0], rt->ev_fork, 0), "pwmlp_train_backward: fork"));
    dst = rt->side[0];
  }
  if (p->dfeat != nullptr)
    CL3D_TRY(cl3d_pwmlp_point_gemm_bwd_data(p->dght, p->wcat, p->B, p->C, p->N, Co, p->precision, p->dfeat, p->gemm_ws_d,
                                            p->gemm_ws_bytes_b, dst));
  if (fork) CL3D_TRY(hip_ok(hipEventRecord(rt->ev_w, dst), "pwmlp_train_backward: event"));
  if (p->dW != nullptr)
    CL3D_TRY(cl3d_pwmlp_point_gemm_bwd_weight(p->features, p->dght, p->dwr, p->B, p->C, p->N, Co, p->precision, p->dW,
                                              p->gemm_ws_w, p->gemm_ws_bytes_b, st));
  if (fork) CL3D_TRY(hip_ok(hipStreamWaitEvent(st, rt->ev_w, 0), "pwmlp_train_backward: join"));
  return CL3D_OK;
}

// ---- the three gather-and-reduce operators (PosPool / AdaptiveWeight / PseudoGrid), one call per pass (round 6) -------
//   forward    caller's stream: ball query ... joins the layout change: the fused reduction; joins the CSR
//                                                     |   side 0: features [B,C,N] -> rows [B,N,C]
//                                                     |   side 1: (behind the query) CSR inverse
//   backward   caller's stream: upstream gradient -> rows, support-major pass, parameter gradients
static int enqueue_reduce_forward(const cl3d_reduce_pass *p, hipStream_t st, cl3d::PassRuntime *rt) {
  using namespace cl3d;
  const bool want_csr = p->inv_off != nullptr && !p->csr_ready;
  CL3D_TRY(hip_ok(hipEventRecord(rt->ev_in, st), "reduce_train_forward: event"));
  if (!p->idx_ready)
    CL3D_TRY(cl3d_masked_ordered_ball_query(p->query_xyz, p->support_xyz, p->query_mask, p->support_mask, p->B, p->M, p->N,
                                            p->radius, p->K, p->idx, p->idx_mask, p->bq_ws, p->bq_ws_bytes, st));
  CL3D_TRY(hip_ok(hipStreamWaitEvent(rt->side[0], rt->ev_in, 0), "reduce_train_forward: fork"));
  CL3D_TRY(cl3d_transpose(p->features, p->B, p->C, p->N, p->ft, rt->side[0]));
  CL3D_TRY(hip_ok(hipEventRecord(rt->ev_fork, rt->side[0]), "reduce_train_forward: event"));
  if (want_csr) {  // (enqueued before the gather pass: the build inherits the query's queue, cl3d_pwmlp_train_forward's note)
    CL3D_REQUIRE(p->inv_slots != nullptr, "reduce_train_forward: null inv_slots");
    CL3D_TRY(hip_ok(hipEventRecord(rt->ev_bq, st), "reduce_train_forward: event"));
    CL3D_TRY(hip_ok(hipStreamWaitEvent(rt->side[1], rt->ev_bq, 0), "reduce_train_forward: fork"));
    CL3D_TRY(cl3d_build_inverse_index(p->idx, p->B, p->N, p->M * p->K, p->inv_off, p->inv_slots, p->csr_ws, p->csr_ws_bytes,
                                      rt->side[1]));
    CL3D_TRY(hip_ok(hipEventRecord(rt->ev_csr, rt->side[1]), "reduce_train_forward: event"));
  }
  CL3D_TRY(hip_ok(hipStreamWaitEvent(st, rt->ev_fork, 0), "reduce_train_forward: join"));
  CL3D_TRY(cl3d_fused_reduce_fwd(p->op, p->query_xyz, p->support_xyz, p->query_mask, p->idx, p->idx_mask, p->ft, p->B, p->N,
                                 p->M, p->K, p->C, p->radius, p->normalize, p->reduction, p->p0, p->p1, p->pint, p->pfloat,
                                 p->constant, p->out, 1, p->slotrec, p->pairs, st));
  if (p->gamma != nullptr)  // the output transform: BatchNorm1d (batch statistics) + ReLU on the raw result
    CL3D_TRY(cl3d_bn_add_relu_train_fwd(p->out, p->gamma, p->beta, p->running_mean, p->running_var, p->num_batches_tracked,
                                        p->eps, p->momentum, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 1,
                                        p->B, p->C, p->M, p->bn_partial, p->bn_parts, p->vec, nullptr, p->act, st));
  if (want_csr) CL3D_TRY(hip_ok(hipStreamWaitEvent(st, rt->ev_csr, 0), "reduce_train_forward: join"));
  return CL3D_OK;
}

static int enqueue_reduce_backward(const cl3d_reduce_pass *p, hipStream_t st) {
  const float *g = p->gout;
  if (p->gamma != nullptr) {  // BatchNorm + ReLU backward: gradient w.r.t. the activated output -> w.r.t. the raw result
    CL3D_TRY(cl3d_bn_add_relu_bwd(p->gout, p->act, p->out, p->vec + 2 * p->C, p->vec + 3 * p->C, p->gamma, nullptr, nullptr,
                                  nullptr, nullptr, 1, p->B, p->C, p->M, (double)p->B * p->M, p->bn_partial, p->bn_parts,
                                  p->coef, nullptr, p->graw, nullptr, st));
    g = p->graw;
  }
  CL3D_TRY(cl3d_transpose(g, p->B, p->C, p->M, p->gout_t, st));
  CL3D_TRY(cl3d_fused_reduce_bwd(p->op, p->gout_t, p->ft, p->slotrec, p->pairs, p->idx, p->inv_off, p->inv_slots, p->B, p->N,
                                 p->M, p->K, p->C, p->p0, p->p1, p->pint, p->pfloat, p->constant, p->dfeat, 1, p->dparam,
                                 p->nparts, st));
  if (p->g1 != nullptr)
    CL3D_TRY(cl3d_fused_param_reduce(p->op, p->dparam, p->nparts, p->C, p->pint, p->g0, p->g1, st));
  return CL3D_OK;
}

extern "C" int cl3d_reduce_train_forward(const cl3d_reduce_pass *p, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(p != nullptr, "reduce_train_forward: null argument block");
  CL3D_REQUIRE(p->B >= 0 && p->N >= 1 && p->M >= 1 && p->K >= 1 && p->C >= 1 && p->radius > 0.f && p->op >= 0 && p->op <= 3,
               "reduce_train_forward: bad sizes");
  CL3D_REQUIRE(p->query_xyz && p->support_xyz && p->query_mask && p->support_mask && p->idx && p->idx_mask && p->features &&
                   p->ft && p->out,
               "reduce_train_forward: null pointer");
  CL3D_REQUIRE(p->gamma == nullptr || (p->beta && p->running_mean && p->running_var && p->act && p->vec && p->bn_partial &&
                                       p->bn_parts == cl3d_bn_partials(p->B, p->C, p->M) && p->momentum >= 0.f),
               "reduce_train_forward: the output transform needs beta, running statistics, act, vec and bn_partial");
  if (p->B == 0) return CL3D_OK;
  PassRuntime *rt = pass_runtime();
  if (rt == nullptr) return fail(CL3D_E_LAUNCH, "reduce_train_forward: side streams could not be created");
  std::lock_guard<std::mutex> use(rt->use);
  return run_pass(0, p, (hipStream_t)stream, rt, rt->dev, [&](hipStream_t s) { return enqueue_reduce_forward(p, s, rt); });
}

extern "C" int cl3d_reduce_train_backward(const cl3d_reduce_pass *p, cl3d_stream_t stream) {
  using namespace cl3d;
  CL3D_REQUIRE(p != nullptr, "reduce_train_backward: null argument block");
  CL3D_REQUIRE(p->gout && p->gout_t && p->ft && p->slotrec && p->idx && p->inv_off && p->inv_slots && p->dfeat,
               "reduce_train_backward: null pointer");
  CL3D_REQUIRE(p->nparts == 0 || p->dparam != nullptr, "reduce_train_backward: null dparam");
  CL3D_REQUIRE(p->gamma == nullptr || (p->act && p->out && p->vec && p->graw && p->coef && p->bn_partial),
               "reduce_train_backward: the output transform needs act, out, vec, graw, coef and bn_partial");
  if (p->B == 0) return CL3D_OK;
  PassRuntime *rt = pass_runtime();
  if (rt == nullptr) return fail(CL3D_E_LAUNCH, "reduce_train_backward: side streams could not be created");
  std::lock_guard<std::mutex> use(rt->use);
  return run_pass(1, p, (hipStream_t)stream, rt, rt->dev, [&](hipStream_t s) { return enqueue_reduce_backward(p, s); });
}

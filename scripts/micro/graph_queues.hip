// graph_queues.hip -- how the HIP runtime lays a captured multi-stream step out on hardware queues.
//
// The bench step is captured from three streams (ball query / CSR build / features) and replayed as one HIP graph; which
// branch shares the launch queue with which decides where the ~10 us cross-queue hand-overs fall (profiles/r04, r05).
// This builds graphs of the step's SHAPE from timed spin kernels (one workgroup each: any two may overlap) in several
// capture orders and prints, for a late replay, every node's start and end relative to the replay's first start:
// ground truth for the layout rules the engine's Python and pass.hip rely on.
//
//   hipcc --offload-arch=gfx950 -O2 -o graph_queues graph_queues.hip && ./graph_queues
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));           \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__global__ void spin(long long ticks, long long *rec, int slot) {
  const long long t0 = wall_clock64();  // 100 MHz
  while (wall_clock64() - t0 < ticks) {
  }
  if (threadIdx.x == 0) {
    rec[2 * slot] = t0;
    rec[2 * slot + 1] = wall_clock64();
  }
}

struct Ctx {
  hipStream_t main, s0, s1, s2;
  long long *rec;
  std::vector<std::string> names;
  int n = 0;
  void k(hipStream_t st, const char *name, double us) {
    names.push_back(name);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, (long long)(us * 100), rec, n++);
  }
  void edge(hipStream_t from, hipStream_t to) {  // `to` waits for what `from` holds now
    hipEvent_t e;
    CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipEventRecord(e, from));
    CK(hipStreamWaitEvent(to, e, 0));
  }
};

typedef void (*Builder)(Ctx &);

// the forward of the step, r4 order: query (side 0), CSR behind it (side 1), product (main), stats pass (main, joins query)
static void fwd(Ctx &c, bool csr_first, bool query_on_main) {
  hipStream_t qs = query_on_main ? c.main : c.s0;
  hipStream_t fs = query_on_main ? c.s0 : c.main;  // where the product runs
  if (!query_on_main) c.edge(c.main, c.s0);
  c.k(qs, "query", 50);
  hipEvent_t eq;
  CK(hipEventCreateWithFlags(&eq, hipEventDisableTiming));
  CK(hipEventRecord(eq, qs));
  auto csr = [&]() {
    CK(hipStreamWaitEvent(c.s1, eq, 0));
    c.k(c.s1, "csr1", 10);
    c.k(c.s1, "csr2", 10);
    c.k(c.s1, "csr3", 10);
  };
  if (csr_first) csr();
  if (query_on_main) c.edge(c.main, c.s0);  // (fork of the product from the capture's origin: no dependency on the query)
  c.k(fs, "weights", 5);
  c.k(fs, "product", 18);
  if (fs != c.main) c.edge(fs, c.main);
  if (qs != c.main) CK(hipStreamWaitEvent(c.main, eq, 0));
  c.k(c.main, "stats", 60);
  if (!csr_first) csr();
  c.k(c.main, "fin0", 5);
  c.k(c.main, "apply", 9);
}
static void bwd(Ctx &c, int tail) {
  c.k(c.main, "rows", 33);
  c.k(c.main, "fin1", 5);
  c.k(c.main, "hit", 10);
  c.edge(c.s1, c.main);  // join the CSR
  c.k(c.main, "support", 50);
  if (tail == 0) {  // one kernel for both gradients
    c.k(c.main, "grads", 27);
    c.k(c.main, "reduce", 8);
  } else {  // r4: weight gradient forked first onto side 2, data gradient on main
    c.edge(c.main, c.s2);
    c.k(c.s2, "wgrad", 37);
    c.k(c.s2, "reduce", 8);
    c.k(c.main, "dgrad", 25);
    c.edge(c.s2, c.main);
  }
  c.k(c.main, "optim", 4);
}

static void g_r4(Ctx &c) { fwd(c, true, false); bwd(c, 1); }
static void g_r4_onegrad(Ctx &c) { fwd(c, true, false); bwd(c, 0); }
static void g_new(Ctx &c) { fwd(c, false, false); bwd(c, 0); }
static void g_new_qmain(Ctx &c) { fwd(c, false, true); bwd(c, 0); }
static void g_r4_qmain(Ctx &c) { fwd(c, true, true); bwd(c, 0); }
// product captured BEFORE the query (root order swapped); the query forks from the capture's origin
static void g_product_first(Ctx &c) {
  hipEvent_t e0, eq;
  CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&eq, hipEventDisableTiming));
  CK(hipEventRecord(e0, c.main));
  c.k(c.main, "weights", 5);
  c.k(c.main, "product", 18);
  CK(hipStreamWaitEvent(c.s0, e0, 0));
  c.k(c.s0, "query", 50);
  CK(hipEventRecord(eq, c.s0));
  CK(hipStreamWaitEvent(c.main, eq, 0));
  c.k(c.main, "stats", 60);
  CK(hipStreamWaitEvent(c.s1, eq, 0));
  c.k(c.s1, "csr1", 10);
  c.k(c.s1, "csr2", 10);
  c.k(c.s1, "csr3", 10);
  c.k(c.main, "fin0", 5);
  c.k(c.main, "apply", 9);
  bwd(c, 0);
}
// the CSR build captured at the END of the forward pass (behind apply), still only waiting for the query
static void g_csr_late(Ctx &c) {
  c.edge(c.main, c.s0);
  c.k(c.s0, "query", 50);
  hipEvent_t eq;
  CK(hipEventCreateWithFlags(&eq, hipEventDisableTiming));
  CK(hipEventRecord(eq, c.s0));
  c.k(c.main, "weights", 5);
  c.k(c.main, "product", 18);
  CK(hipStreamWaitEvent(c.main, eq, 0));
  c.k(c.main, "stats", 60);
  c.k(c.main, "fin0", 5);
  c.k(c.main, "apply", 9);
  CK(hipStreamWaitEvent(c.s1, eq, 0));
  c.k(c.s1, "csr1", 10);
  c.k(c.s1, "csr2", 10);
  c.k(c.s1, "csr3", 10);
  bwd(c, 0);
}
// A: query first (root 1), stats its first dependent, CSR its second AND made to wait for the product as well (a
// dependency it does not need: the product ends long before the query) -- so that the side queue's order must be
// weights, product, csr
static void g_fake_dep(Ctx &c) {
  hipEvent_t e0, eq, ep;
  CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&eq, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ep, hipEventDisableTiming));
  CK(hipEventRecord(e0, c.main));
  CK(hipStreamWaitEvent(c.s0, e0, 0));
  c.k(c.s0, "query", 50);
  CK(hipEventRecord(eq, c.s0));
  c.k(c.main, "weights", 5);
  c.k(c.main, "product", 18);
  CK(hipEventRecord(ep, c.main));
  CK(hipStreamWaitEvent(c.main, eq, 0));
  c.k(c.main, "stats", 60);
  CK(hipStreamWaitEvent(c.s1, eq, 0));
  CK(hipStreamWaitEvent(c.s1, ep, 0));
  c.k(c.s1, "csr1", 10);
  c.k(c.s1, "csr2", 10);
  c.k(c.s1, "csr3", 10);
  c.k(c.main, "fin0", 5);
  c.k(c.main, "apply", 9);
  bwd(c, 0);
}
// B: product captured first (root 1: the whole feature chain follows it), query forked from the origin with the CSR
// right behind it (its first dependent), stats joins the query
static void g_product_root(Ctx &c) {
  hipEvent_t e0, eq;
  CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&eq, hipEventDisableTiming));
  CK(hipEventRecord(e0, c.main));
  c.k(c.main, "weights", 5);
  c.k(c.main, "product", 18);
  CK(hipStreamWaitEvent(c.s0, e0, 0));
  c.k(c.s0, "query", 50);
  CK(hipEventRecord(eq, c.s0));
  CK(hipStreamWaitEvent(c.s1, eq, 0));
  c.k(c.s1, "csr1", 10);
  c.k(c.s1, "csr2", 10);
  c.k(c.s1, "csr3", 10);
  CK(hipStreamWaitEvent(c.main, eq, 0));
  c.k(c.main, "stats", 60);
  c.k(c.main, "fin0", 5);
  c.k(c.main, "apply", 9);
  bwd(c, 0);
}
// C: as A, but the weights kernel is the graph's ONLY root (the query waits for it: 5 us): no second root to place
static void g_single_root(Ctx &c) {
  hipEvent_t ew, eq, ep;
  CK(hipEventCreateWithFlags(&ew, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&eq, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ep, hipEventDisableTiming));
  c.k(c.main, "weights", 5);
  CK(hipEventRecord(ew, c.main));
  CK(hipStreamWaitEvent(c.s0, ew, 0));
  c.k(c.s0, "query", 50);     // weights' first dependent
  CK(hipEventRecord(eq, c.s0));
  c.k(c.main, "product", 18);  // weights' second dependent
  CK(hipEventRecord(ep, c.main));
  CK(hipStreamWaitEvent(c.main, eq, 0));
  c.k(c.main, "stats", 60);    // query's first dependent
  CK(hipStreamWaitEvent(c.s1, eq, 0));
  CK(hipStreamWaitEvent(c.s1, ep, 0));
  c.k(c.s1, "csr1", 10);       // query's second, product's second
  c.k(c.s1, "csr2", 10);
  c.k(c.s1, "csr3", 10);
  c.k(c.main, "fin0", 5);
  c.k(c.main, "apply", 9);
  bwd(c, 0);
}
// everything on one stream (no overlap): the floor of the hand-over costs
static void g_serial(Ctx &c) {
  c.k(c.main, "query", 50);
  c.k(c.main, "weights", 5);
  c.k(c.main, "product", 18);
  c.k(c.main, "stats", 60);
  c.k(c.main, "csr1", 10);
  c.k(c.main, "csr2", 10);
  c.k(c.main, "csr3", 10);
  c.k(c.main, "fin0", 5);
  c.k(c.main, "apply", 9);
  c.k(c.main, "rows", 33);
  c.k(c.main, "fin1", 5);
  c.k(c.main, "hit", 10);
  c.k(c.main, "support", 50);
  c.k(c.main, "grads", 27);
  c.k(c.main, "reduce", 8);
  c.k(c.main, "optim", 4);
}

static void run(const char *title, Builder b) {
  Ctx c;
  CK(hipStreamCreateWithFlags(&c.main, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&c.s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&c.s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&c.s2, hipStreamNonBlocking));
  CK(hipMalloc(&c.rec, 64 * 2 * sizeof(long long)));
  hipGraph_t g;
  hipGraphExec_t ex;
  CK(hipStreamBeginCapture(c.main, hipStreamCaptureModeGlobal));
  b(c);
  CK(hipStreamEndCapture(c.main, &g));
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ex, c.main));
  CK(hipStreamSynchronize(c.main));
  std::vector<long long> h(2 * c.n);
  CK(hipMemcpy(h.data(), c.rec, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  long long t0 = h[0], t1 = h[1];
  for (int i = 0; i < c.n; ++i) {
    t0 = h[2 * i] < t0 ? h[2 * i] : t0;
    t1 = h[2 * i + 1] > t1 ? h[2 * i + 1] : t1;
  }
  printf("== %s: %d nodes, replay makespan %.1f us\n", title, c.n, (t1 - t0) / 100.0);
  for (int i = 0; i < c.n; ++i)
    printf("   %-8s start %7.1f  end %7.1f\n", c.names[i].c_str(), (h[2 * i] - t0) / 100.0, (h[2 * i + 1] - t0) / 100.0);
  // steady-state period over 50 replays
  hipEvent_t a, z;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&z));
  CK(hipEventRecord(a, c.main));
  for (int r = 0; r < 50; ++r) CK(hipGraphLaunch(ex, c.main));
  CK(hipEventRecord(z, c.main));
  CK(hipEventSynchronize(z));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, z));
  printf("   period %.1f us per replay (critical path of the spins: 50+60+5+9+33+5+10+50+27+8+4 = 261)\n", ms * 1e3 / 50);
  CK(hipGraphExecDestroy(ex));
  CK(hipGraphDestroy(g));
}

int main() {
  run("r4 capture order, forked gradient pair", g_r4);
  run("r4 capture order, one gradient kernel", g_r4_onegrad);
  run("CSR captured behind the stats pass", g_new);
  run("query on the capture's own stream, product forked, CSR behind the stats pass", g_new_qmain);
  run("query on the capture's own stream, product forked, CSR first", g_r4_qmain);
  run("product captured first, query forked from the origin, CSR behind the stats pass", g_product_first);
  run("CSR captured at the end of the forward pass", g_csr_late);
  run("one stream", g_serial);
  run("A: query root, stats first, CSR second + waits for the product", g_fake_dep);
  run("B: product root, query forked from the origin, CSR first behind the query", g_product_root);
  run("C: weights the only root; query its first dependent, product its second; CSR waits for query and product", g_single_root);
  return 0;
}

// ctc_step.cpp -- the benchmark step of benchmarks/ctc.cpp:136-168 written against
// the drop-in C++ API (include/gtn/), i.e. what a gtn user's host code looks like
// on this engine: target graphs are built on host threads with parallelMap, the
// graph functions run batched on the GPU through their vector overloads.
// Built into bench_native/libgtn_bench.so and driven by bench.py over ctypes.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "gtn/gtn.h"

#include "../gtn_amd/criteria/ctc_criterion.h"

using namespace gtn;

// host milliseconds of the last step's phases on the calling thread: target graphs (parallelMap), emissions
// graphs, intersect, the two forwardScores + subtract, backward -- what bounds a step once the GPU work is short
static double g_last[5] = {0, 0, 0, 0, 0};
static std::string g_error;
extern "C" __attribute__((visibility("default"))) const char* gtn_bench_last_error() { return g_error.c_str(); }
extern "C" __attribute__((visibility("default"))) void gtn_bench_last_host_ms(double* out5) {
  for (int i = 0; i < 5; ++i) out5[i] = g_last[i];
}

// emissions: DEVICE [B][T][C]; targets: host [B][U]; loss_dev: DEVICE [B];
// grad_dev: DEVICE [B][T][C] or null.  Returns 0 or a gtnx status.
extern "C" __attribute__((visibility("default"))) int gtn_bench_ctc_step(const void* emissions, const int* targets,
                                                                         int B, int T, int C, int U, void* loss_dev,
                                                                         void* grad_dev) {
  static const bool timing = std::getenv("GTN_BENCH_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  auto tq = now();
  static auto t_prev_end = tq;
  decltype(tq) t_tail0 = tq, t_tail1 = tq;
  int rc = 0;
  try {
    std::vector<int> len(B, U);
    // fwd + bwd of benchmarks/ctc.cpp:150-165, batched (gtn_amd/criteria/ctc_criterion.h)
    criteria::CtcStepTimes tm;
    t_tail0 = now();
    criteria::ctcLossBatch(emissions, targets, len.data(), B, T, C, /*blank=*/0, loss_dev, grad_dev, /*targetGrad=*/true, &tm);
    g_last[0] = tm.build;
    g_last[1] = tm.linear;
    g_last[2] = tm.intersect;
    g_last[3] = tm.forward;
    g_last[4] = tm.backward;
    if (timing)
      std::fprintf(stderr, "host ms: build %.2f linear %.2f intersect %.2f fwd %.2f bwd %.2f\n", tm.build, tm.linear,
                   tm.intersect, tm.forward, tm.backward);
    t_tail1 = now();
  } catch (const std::exception& e) {
    g_error = e.what();
    rc = -1;
  }
  auto t_end = now();
  if (timing)
    std::fprintf(stderr, "host ms: copies %.2f teardown %.2f  between calls %.2f\n", ms(t_tail0, t_tail1),
                 ms(t_tail1, t_end), ms(t_prev_end, tq));
  t_prev_end = now();
  return rc;
}

// ctc_step.cpp -- the benchmark step of benchmarks/ctc.cpp:136-168 written against
// the drop-in C++ API (include/gtn/), i.e. what a gtn user's host code looks like
// on this engine: target graphs are built on host threads with parallelMap, the
// graph functions run batched on the GPU through their vector overloads.
// Built into bench_native/libgtn_bench.so and driven by bench.py over ctypes.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gtn/gtn.h"

using namespace gtn;

namespace {
// benchmarks/ctc.cpp:40-58
Graph ctcGraph(const std::vector<int>& target) {
  int blank = 0;
  size_t L = 2 * target.size() + 1;
  Graph ctc;
  for (size_t l = 0; l < L; l++) {
    size_t idx = (l - 1) / 2;
    ctc.addNode(l == 0, l == L - 1 || l == L - 2);
    int label = l % 2 ? target[idx] : blank;
    ctc.addArc(l, l, label);
    if (l > 0) ctc.addArc(l - 1, l, label);
    if (l % 2 && l > 1 && label != target[idx - 1]) ctc.addArc(l - 2, l, label);
  }
  ctc.arcSort();
  return ctc;
}
} // namespace

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
    auto t0 = now();
    std::vector<std::vector<int>> tg(B);
    for (int b = 0; b < B; ++b) tg[b].assign(targets + (size_t)b * U, targets + (size_t)(b + 1) * U);
    // fwd of benchmarks/ctc.cpp:150-158, batched
    auto ctcs = parallelMap(ctcGraph, tg);
    auto t1 = now();
    auto ems = linearGraphs(B, T, C, emissions);  // linearGraph + setWeights, one copy
    auto t2 = now();
    auto comp = batched::intersect(ctcs, ems);
    auto t3 = now();
    // (named in this order: C++ leaves the evaluation order of call arguments open)
    auto norm = batched::forwardScore(ems);
    auto score = batched::forwardScore(comp);
    auto losses = batched::subtract(norm, score);
    auto t4 = now();
    // bwd of benchmarks/ctc.cpp:160
    batched::backward(losses);
    auto t5 = now();
    if (timing)
      std::fprintf(stderr, "host ms: build %.2f linear %.2f intersect %.2f fwd %.2f bwd %.2f\n", ms(t0, t1), ms(t1, t2),
                   ms(t2, t3), ms(t3, t4), ms(t4, t5));
    t_tail0 = now();
    auto h = detail::handles(losses);
    detail::check(gtnx_items_device_n(h.data(), B, loss_dev));
    if (grad_dev) {
      auto he = detail::handles(ems);
      std::vector<int64_t> off(B);
      for (int b = 0; b < B; ++b) off[b] = (int64_t)b * T * C;
      detail::check(gtnx_grads_device_n(he.data(), B, grad_dev, off.data()));
    }
    t_tail1 = now();
  } catch (const std::exception& e) {
    rc = -1;
  }
  auto t_end = now();
  if (timing)
    std::fprintf(stderr, "host ms: copies %.2f teardown %.2f  between calls %.2f\n", ms(t_tail0, t_tail1),
                 ms(t_tail1, t_end), ms(t_prev_end, tq));
  t_prev_end = now();
  return rc;
}

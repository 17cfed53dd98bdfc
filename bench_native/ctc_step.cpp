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
static bool g_vector_symbolic = true;
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

// The same step through the VECTOR overloads of the per-graph functions (gtn::batched, the device analogue of the
// binding's std::vector<Graph> forms, bindings/python/gtn/_functions.cpp:84-135): B target graphs built on host
// threads, B emission graphs over the device tensor, one batched launch per function, B result graphs per call.
// What a caller gets who batches by hand but keeps per-utterance graphs; bench.py reports it next to the batch
// records above and the reference's own loop (tests/native/bm_ctc_c256.cpp).
extern "C" __attribute__((visibility("default"))) int gtn_bench_ctc_step_vector(const void* emissions, const int* targets,
                                                                                int B, int T, int C, int U,
                                                                                void* loss_dev, void* grad_dev) {
  static const bool timing = std::getenv("GTN_BENCH_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  try {
    const auto t0 = now();
    std::vector<std::vector<int>> tg(static_cast<size_t>(B));
    for (int b = 0; b < B; ++b) tg[b].assign(targets + size_t(b) * U, targets + size_t(b + 1) * U);
    auto ctcs = parallelMap([](const std::vector<int>& t) { return criteria::ctcTargetGraph(t); }, tg);
    const auto t1 = now();
    auto ems = linearGraphs(B, T, C, emissions);
    std::vector<gtnx_graph_t> he(static_cast<size_t>(B));
    std::vector<int64_t> off(static_cast<size_t>(B));
    for (int b = 0; b < B; ++b) {
      he[b] = ems[b].handle();
      off[b] = int64_t(b) * T * C;
    }
    if (grad_dev) detail::check(gtnx_grads_bind_device_n(he.data(), B, grad_dev, off.data()));
    const auto t2 = now();
    double t_int, t_fs, t_bw, t_out;
    {
      SymbolicCompose symbolic(g_vector_symbolic ? 2 : 0);  // (2: only forwardScore of the lattices is taken)
      auto comp = batched::intersect(ctcs, ems);
      const auto t3 = now();
      auto score = batched::forwardScore(comp);
      auto norm = batched::forwardScore(ems);
      auto losses = batched::subtract(norm, score);
      const auto t4 = now();
      if (grad_dev) batched::backward(losses);
      const auto t5 = now();
      std::vector<gtnx_graph_t> hl(static_cast<size_t>(B));
      for (int b = 0; b < B; ++b) hl[b] = losses[b].handle();
      detail::check(gtnx_items_device_n(hl.data(), B, loss_dev));
      if (grad_dev) detail::check(gtnx_grads_device_n(he.data(), B, grad_dev, off.data()));
      const auto t6 = now();
      t_int = ms(t2, t3), t_fs = ms(t3, t4), t_bw = ms(t4, t5), t_out = ms(t5, t6);
    }
    const auto t7 = now();
    if (timing)
      std::fprintf(stderr, "vector step host ms: targets %.2f emissions %.2f intersect %.2f scores %.2f backward %.2f outputs %.2f\n",
                   ms(t0, t1), ms(t1, t2), t_int, t_fs, t_bw, t_out);
    (void)t7;
    return 0;
  } catch (const std::exception& e) {
    g_error = e.what();
    return -1;
  }
}
// 0: the lattices of the vector step are BUILT (compose_kernel, the forwardScore kernel over the built lattice and its
// fused backward: bench.py's built_lattice_path); 1 (default): kept symbolic
extern "C" __attribute__((visibility("default"))) void gtn_bench_vector_symbolic(int on) { g_vector_symbolic = on != 0; }

// BASELINE config C1 (benchmarks/ctc.cpp with batch = 1: T = 100, alphabet 28, U = 20): ONE utterance through the
// per-graph functions, reference names only, nothing batched, no compose-mode hint -- so the engine's default
// policy applies (gtn_amd.h: gtnx_compose_mode -1): the product of a target built on the host and the emissions
// stays symbolic and is swept by band.hip (GTNX_LAZY_COMPOSE=0 builds it: compose_kernel, the forwardScore kernel
// over the lattice, its fused backward -- 0.64 ms per loss against 0.19); `iters` repetitions, each ended by reading
// the loss (item()).  Returns the mean milliseconds per repetition (< 0: error).  emissions: DEVICE [T][C].
extern "C" __attribute__((visibility("default"))) double gtn_bench_single_utterance(const void* emissions, const int* target,
                                                                                  int T, int C, int U, int iters,
                                                                                  float* loss_out) {
  try {
    std::vector<int> tg(target, target + U);
    auto once = [&]() {
      Graph ctc = criteria::ctcTargetGraph(tg);  // (the graph of benchmarks/ctc.cpp:40-58)
      Graph em = linearGraph(T, C);
      em.setWeights(static_cast<const float*>(emissions));
      Graph loss = subtract(forwardScore(em), forwardScore(intersect(ctc, em)));
      backward(loss);
      return loss.item();
    };
    float l = 0;
    for (int i = 0; i < 10; ++i) l = once();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) l = once();
    const auto t1 = std::chrono::steady_clock::now();
    if (loss_out) *loss_out = l;
    return std::chrono::duration<double, std::milli>(t1 - t0).count() / iters;
  } catch (const std::exception& e) {
    g_error = e.what();
    return -1.0;
  }
}

// BASELINE config C2 (forwardScore on 256 linear-chain emission graphs, T = 150, C = 32) from C++: the graphs over
// the device tensor (linearGraphs = linearGraph + setWeights for the batch), the vector overload of forwardScore --
// one batched launch -- and the scores left in a device tensor; `iters` repetitions after 10 of warm-up, ended by
// one synchronisation.  Returns the mean milliseconds per batch (< 0: error).  emissions: DEVICE [B][T][C].
extern "C" __attribute__((visibility("default"))) double gtn_bench_forward_score_linear(const void* emissions, int B, int T,
                                                                                      int C, void* scores_dev, int iters) {
  try {
    std::vector<gtnx_graph_t> hs(static_cast<size_t>(B));
    auto once = [&]() {
      auto ems = linearGraphs(B, T, C, emissions);
      auto scores = batched::forwardScore(ems);
      for (int b = 0; b < B; ++b) hs[size_t(b)] = scores[size_t(b)].handle();
      detail::check(gtnx_items_device_n(hs.data(), B, scores_dev));
    };
    for (int i = 0; i < 10; ++i) once();
    detail::check(gtnx_synchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) once();
    detail::check(gtnx_synchronize());
    const auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(t1 - t0).count() / iters;
  } catch (const std::exception& e) {
    g_error = e.what();
    return -1.0;
  }
}

// The same through gtn::Batch (batch records: the B chains over the tensor as ONE object, include/gtn/batch.h): the
// form bench.py's headline uses for C3.  Returns the mean milliseconds per batch (< 0: error).
extern "C" __attribute__((visibility("default"))) double gtn_bench_forward_score_linear_batch(const void* emissions, int B, int T,
                                                                                            int C, void* scores_dev, int iters) {
  try {
    auto once = [&]() {
      Batch ems = Batch::linear(B, T, C, emissions, /*calcGrad=*/true, /*borrow=*/true);
      batched::forwardScore(ems).itemsToDevice(scores_dev);
    };
    for (int i = 0; i < 10; ++i) once();
    detail::check(gtnx_synchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) once();
    detail::check(gtnx_synchronize());
    const auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(t1 - t0).count() / iters;
  } catch (const std::exception& e) {
    g_error = e.what();
    return -1.0;
  }
}

// BASELINE config C3's shape through the reference's decode: parallelMap over a per-utterance function that builds
// the target graph and the emission graph and returns viterbiPath(intersect(ctc, emissions)) (functions.cpp:324-330;
// the loop of benchmarks/ctc.cpp with viterbiPath in place of the loss), reference names only.  `iters` repetitions
// after 2 of warm-up, each ended by looking at every path (numArcs).  Returns the mean milliseconds per batch
// (< 0: error); the output labels of the last repetition's paths go to labels_out ([B][T], -1 padded) when given.
// emissions: DEVICE [B][T][C]; targets: host [B][U].
extern "C" __attribute__((visibility("default"))) double gtn_bench_viterbi_reference_loop(const void* emissions,
                                                                                        const int* targets, int B, int T,
                                                                                        int C, int U, int iters,
                                                                                        int* labels_out) {
  try {
    std::vector<std::vector<int>> tg(static_cast<size_t>(B));
    std::vector<const float*> scores(static_cast<size_t>(B));
    for (int b = 0; b < B; ++b) {
      tg[b].assign(targets + size_t(b) * U, targets + size_t(b + 1) * U);
      scores[b] = static_cast<const float*>(emissions) + size_t(b) * T * C;
    }
    auto decode = [T, C](const std::vector<int>& target, const float* emissionsScore) {
      Graph ctc = criteria::ctcTargetGraph(target);
      Graph emissions = linearGraph(T, C);
      emissions.setWeights(emissionsScore);
      return viterbiPath(intersect(ctc, emissions));
    };
    std::vector<Graph> paths;
    size_t arcs = 0;
    auto once = [&]() {
      paths = parallelMap(decode, tg, scores);
      for (auto& p : paths) arcs += size_t(p.numArcs());
    };
    for (int i = 0; i < 2; ++i) once();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) once();
    const auto t1 = std::chrono::steady_clock::now();
    if (labels_out) {
      for (int b = 0; b < B; ++b) {
        const int n = int(paths[b].numArcs());
        for (int t = 0; t < T; ++t) labels_out[size_t(b) * T + t] = t < n ? paths[b].olabel(t) : -1;
      }
    }
    (void)arcs;
    return std::chrono::duration<double, std::milli>(t1 - t0).count() / iters;
  } catch (const std::exception& e) {
    g_error = e.what();
    return -1.0;
  }
}

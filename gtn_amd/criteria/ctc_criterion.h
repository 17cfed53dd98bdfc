// ctc_criterion.h -- the CTC criterion of benchmarks/ctc.cpp:40-58,150-165 (same as
// examples/ctc.cpp:21-41 and bindings/python/examples/pytorch_loss.py) written against the
// drop-in C++ API of include/gtn, batched: host threads build the target graphs, the graph
// functions run as batch-of-graphs launches on the GPU, losses and emission gradients stay
// on the device.  Header-only; used by bench_native/ctc_step.cpp and by
// gtn_amd/criteria/criteria_capi.cpp (the PyTorch loss).
#pragma once

#include <chrono>
#include <cstdint>
#include <vector>

#include "gtn/gtn.h"

namespace gtn {
namespace criteria {

/** target acceptor: 2U+1 states, blank at even states, skip arcs between different labels */
inline Graph ctcTargetGraph(const std::vector<int>& target, int blank = 0, bool calcGrad = true) {
  // the graph of benchmarks/ctc.cpp:40-58, handed to the engine in two bulk calls (same node and
  // arc order as the reference's addNode / addArc loop)
  const int L = 2 * (int)target.size() + 1;
  std::vector<uint8_t> st(L, 0), ac(L, 0);
  std::vector<int> src, dst, lab;
  src.reserve(3 * L);
  dst.reserve(3 * L);
  lab.reserve(3 * L);
  for (int l = 0; l < L; l++) {
    const int idx = (l - 1) / 2;
    st[l] = l == 0;
    ac[l] = l == L - 1 || l + 2 == L;
    const int label = l % 2 ? target[idx] : blank;
    src.push_back(l), dst.push_back(l), lab.push_back(label);
    if (l > 0) src.push_back(l - 1), dst.push_back(l), lab.push_back(label);
    if (l % 2 && l > 1 && label != target[idx - 1]) src.push_back(l - 2), dst.push_back(l), lab.push_back(label);
  }
  Graph ctc(calcGrad);
  detail::check(gtnx_graph_add_nodes(ctc.handle(), L, st.data(), ac.data()));
  detail::check(gtnx_graph_add_arcs(ctc.handle(), (int)src.size(), src.data(), dst.data(), lab.data(), lab.data(), nullptr));
  ctc.arcSort();
  return ctc;
}

struct CtcStepTimes {
  double build = 0, linear = 0, intersect = 0, forward = 0, backward = 0;
};

/** forward + backward of  loss_b = forwardScore(emissions_b) - forwardScore(target_b ∩ emissions_b)
 *  for a batch -- benchmarks/ctc.cpp:150-165 with one Batch per call instead of parallelMap over
 *  per-utterance graphs.  `emissions`: device [B][T][C], read in place; `lossDev`: device [B];
 *  `gradDev`: device [B][T][C] (d loss / d emissions, written in place) or null.  Targets may have
 *  different lengths. */
inline void ctcLossBatch(
    const void* emissions,
    const int* labels,   // target sequences back to back
    const int* lengths,  // [B]
    int B,
    int T,
    int C,
    int blank,
    void* lossDev,
    void* gradDev,
    bool targetGrad = true,  // benchmarks/ctc.cpp builds its targets with calcGrad = true
    CtcStepTimes* times = nullptr,
    Batch* targetsOut = nullptr) {  // the target acceptors (their gradients populated when targetGrad) handed back
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  auto t0 = now();
  Batch ctcs = Batch::ctcTargets(labels, lengths, B, blank, targetGrad);
  auto t1 = now();
  Batch ems = Batch::linear(B, T, C, emissions, gradDev != nullptr, /*borrow=*/true);
  std::vector<int64_t> off(B);
  for (int b = 0; b < B; ++b) off[b] = (int64_t)b * T * C;
  if (gradDev) ems.bindGrads(gradDev, off.data());
  auto t2 = now();
  // only forwardScore of the lattices is taken: they are never built (band.hip sweeps them)
  Batch comp = batched::intersect(ctcs, ems);
  auto t3 = now();
  // (C++ leaves the evaluation order of benchmarks/ctc.cpp:157's call arguments open; this order lets the
  //  sweep over target o emissions, which reads every emission anyway, leave forwardScore(emissions) behind)
  Batch score = batched::forwardScore(comp);
  Batch norm = batched::forwardScore(ems);
  Batch losses = batched::subtract(norm, score, lossDev);  // (written where the caller wants them: no copy below)
  auto t4 = now();
  if (gradDev) batched::backward(losses);
  auto t5 = now();
  if (times) *times = {ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5)};
  losses.itemsToDevice(lossDev);
  if (gradDev) ems.gradsToDevice(gradDev, off.data());  // (nothing to copy when the rows were written in place)
  if (targetsOut) *targetsOut = std::move(ctcs);
}

inline void ctcLossBatch(
    const void* emissions,
    const std::vector<std::vector<int>>& targets,
    int T,
    int C,
    int blank,
    void* lossDev,
    void* gradDev,
    bool targetGrad = true,
    CtcStepTimes* times = nullptr) {
  std::vector<int> flat, len;
  for (auto& t : targets) {
    flat.insert(flat.end(), t.begin(), t.end());
    len.push_back((int)t.size());
  }
  ctcLossBatch(emissions, flat.data(), len.data(), (int)targets.size(), T, C, blank, lossDev, gradDev, targetGrad, times);
}

} // namespace criteria
} // namespace gtn

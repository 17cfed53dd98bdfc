// gather_test.cpp -- the caller pattern of benchmarks/ctc.cpp:150-165 (per-utterance graph functions
// from parallelMap threads) on seeded inputs, checked against the same functions called one utterance
// at a time on the main thread: the engine gathers the threads' calls into batched launches
// (gtnx_parallel_enter) and keeps the lattices symbolic, and neither may change a result.
// Own test program (not reference code); built by tests/dropin/Makefile, run by tests/test_dropin_gpu.py.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "gtn/gtn.h"

using namespace gtn;

static Graph ctcGraph(const std::vector<int>& target) {
  const int blank = 0;
  const int L = 2 * (int)target.size() + 1;
  Graph ctc;
  for (int l = 0; l < L; l++) {
    const int idx = (l - 1) / 2;
    ctc.addNode(l == 0, l == L - 1 || l == L - 2);
    const int label = l % 2 ? target[idx] : blank;
    ctc.addArc(l, l, label);
    if (l > 0) ctc.addArc(l - 1, l, label);
    if (l % 2 && l > 1 && label != target[idx - 1]) ctc.addArc(l - 2, l, label);
  }
  ctc.arcSort();
  return ctc;
}

int main() {
  const int B = 96, T = 120, M = 20;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> ud(-5.f, 5.f);
  std::vector<std::vector<int>> targets(B);
  std::vector<std::vector<float>> scores(B);
  for (int b = 0; b < B; ++b) {
    const int U = 1 + int(rng() % 30);
    for (int u = 0; u < U; ++u) targets[b].push_back(1 + int(rng() % (M - 1)));
    scores[b].resize(size_t(T) * M);
    for (auto& v : scores[b]) v = ud(rng);
  }
  std::vector<Graph> ems(B), ctcs(B);
  auto fwd = [&](int b) {
    ctcs[b] = ctcGraph(targets[b]);
    ems[b] = linearGraph(T, M);
    ems[b].setWeights(scores[b].data());
    return subtract(forwardScore(ems[b]), forwardScore(intersect(ctcs[b], ems[b])));
  };
  auto bwd = [](const Graph& g) { backward(g); };
  std::vector<int> idx(B);
  for (int b = 0; b < B; ++b) idx[b] = b;
  // gathered: from parallelMap threads
  auto losses = parallelMap(fwd, idx);
  parallelMap(bwd, losses);
  std::vector<float> loss_g(B);
  std::vector<std::vector<float>> eg(B), tg(B);
  for (int b = 0; b < B; ++b) {
    loss_g[b] = losses[b].item();
    eg[b].assign(ems[b].grad().weights(), ems[b].grad().weights() + size_t(T) * M);
    tg[b].assign(ctcs[b].grad().weights(), ctcs[b].grad().weights() + ctcs[b].numArcs());
  }
  // one utterance at a time on this thread
  int bad = 0;
  double worst = 0;
  for (int b = 0; b < B; ++b) {
    Graph l = fwd(b);
    backward(l);
    const float want = l.item();
    if (std::fabs(want - loss_g[b]) > 1e-4f * std::fmax(1.f, std::fabs(want))) ++bad;
    const float* g = ems[b].grad().weights();
    for (size_t i = 0; i < size_t(T) * M; ++i) worst = std::fmax(worst, std::fabs(double(g[i]) - eg[b][i]));
    // (an arc of the target collects a posterior per frame: its gradient can be of the order of T)
    const float* t = ctcs[b].grad().weights();
    for (size_t i = 0; i < tg[b].size(); ++i)
      worst = std::fmax(worst, std::fabs(double(t[i]) - tg[b][i]) / std::fmax(1.0, std::fabs(double(t[i]))));
  }
  std::printf("gathered vs one-by-one: %d of %d losses differ, largest (relative) gradient difference %.3g\n", bad, B, worst);
  if (bad || !(worst < 2e-4)) return 1;
  std::printf("All tests passed\n");
  return 0;
}

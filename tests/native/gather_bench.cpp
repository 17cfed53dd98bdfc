// gather_bench.cpp -- where a batch of the caller pattern of benchmarks/ctc.cpp:136-168 spends its wall time
// (diagnostic; own program, same shape as timeBatchedCtc: B utterances, T = 1000, U = 100, alphabet 28):
// target graphs / emission graphs / the gathered graph functions / backward / letting go of the step's graphs.
//   g++ -O1 -std=c++17 -I include tests/native/gather_bench.cpp -L gtn_amd/lib -lgtn_amd -pthread
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "gtn/gtn.h"

using namespace gtn;
using Clock = std::chrono::steady_clock;
static double ms(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

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
  return ctc;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 512, T = 1000, U = 100, M = 28, iters = 8;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> ud(-5.f, 5.f);
  std::vector<std::vector<int>> targets(B);
  std::vector<std::vector<float>> scores(B);
  for (int b = 0; b < B; ++b) {
    for (int u = 0; u < U; ++u) targets[b].push_back(1 + int(rng() % (M - 1)));
    scores[b].resize(size_t(T) * M);
    for (auto& v : scores[b]) v = ud(rng);
  }
  double t_tg = 0, t_em = 0, t_fn = 0, t_bw = 0, t_free = 0, t_all = 0;
  for (int it = 0; it < iters + 2; ++it) {
    const auto a0 = Clock::now();
    auto tg = parallelMap([](const std::vector<int>& t) { return ctcGraph(t); }, targets);
    const auto a1 = Clock::now();
    auto em = parallelMap(
        [T, M](const std::vector<float>& s) {
          auto e = linearGraph(T, M);
          e.setWeights(s.data());
          return e;
        },
        scores);
    const auto a2 = Clock::now();
    auto losses = parallelMap([](const Graph& c, const Graph& e) { return subtract(forwardScore(e), forwardScore(intersect(c, e))); }, tg, em);
    const auto a3 = Clock::now();
    parallelMap([](const Graph& g) { backward(g); }, losses);
    const float first = losses[0].item();  // (waits for the device)
    const auto a4 = Clock::now();
    losses.clear();
    tg.clear();
    em.clear();
    gtnx_reclaim();
    const auto a5 = Clock::now();
    if (it >= 2) {
      t_tg += ms(a0, a1), t_em += ms(a1, a2), t_fn += ms(a2, a3), t_bw += ms(a3, a4), t_free += ms(a4, a5), t_all += ms(a0, a5);
    }
    if (it == 0) std::printf("loss[0] %.4f\n", first);
  }
  std::printf("B %d per batch [ms]: targets %.2f  emissions %.2f  functions %.2f  backward+wait %.2f  release %.2f  | total %.2f\n", B,
              t_tg / iters, t_em / iters, t_fn / iters, t_bw / iters, t_free / iters, t_all / iters);
  return 0;
}

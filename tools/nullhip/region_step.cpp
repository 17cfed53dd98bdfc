// region_step.cpp -- the reference's caller pattern (benchmarks/ctc.cpp:136-168: parallelMap(fwd), parallelMap(bwd))
// against the null HIP device: host milliseconds per step and, with GTN_HOST_SAMPLE=<file>, stack samples of the
// main thread (the one that joins the regions and runs the deferred calls).  Diagnostic only.
//   make -C tools/nullhip && LD_PRELOAD=tools/nullhip/_bin/libnullhip.so tools/nullhip/_bin/region_step [steps] [B] [C]
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdlib>
#include <vector>

#include "gtn/gtn.h"
#include "sampler.h"

using namespace gtn;

static Graph ctcGraph(const std::vector<int>& target) {
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

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 50, B = argc > 2 ? atoi(argv[2]) : 512, M = argc > 3 ? atoi(argv[3]) : 256;
  const int T = 1000, U = 100;
  std::vector<std::vector<int>> targets(B);
  for (int b = 0; b < B; ++b)
    for (int u = 0; u < U; ++u) targets[b].push_back(1 + int((unsigned(b * 131 + u) * 2654435761u >> 7) % unsigned(M - 1)));
  float* dev = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&dev), sizeof(float) * size_t(B) * T * M) != hipSuccess) return 2;
  std::vector<const float*> scores(B);
  for (int b = 0; b < B; ++b) scores[b] = dev + size_t(b) * T * M;
  auto fwd = [T, M](const std::vector<int>& target, const float* e) {
    auto ctc = ctcGraph(target);
    auto emissions = linearGraph(T, M);
    emissions.setWeights(e);
    return subtract(forwardScore(emissions), forwardScore(intersect(ctc, emissions)));
  };
  auto bwd = [](const Graph& g) { backward(g); };
  const char* sample = std::getenv("GTN_HOST_SAMPLE");
  double sum = 0;
  for (int s = 0; s < steps; ++s) {
    if (s == 3 && sample) start_sampler();
    const auto t0 = std::chrono::steady_clock::now();
    {
      auto lossGraphs = parallelMap(fwd, targets, scores);
      parallelMap(bwd, lossGraphs);
      (void)lossGraphs.back().item();
    }
    if (s >= 3) sum += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  if (sample) dump_samples(sample);
  std::printf("host ms/step: mean %.3f (B=%d C=%d)\n", sum / (steps - 3), B, M);
  return 0;
}

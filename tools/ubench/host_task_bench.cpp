// host_task_bench.cpp -- what one task of the reference's fwd lambda (benchmarks/ctc.cpp:150-158) costs the
// host thread that runs it inside a parallelMap region, piece by piece, and what an empty parallelMap costs
// (pool wake-up + join) for several pool sizes.  Diagnostic; run on the GPU box:
//   g++ -O2 -std=c++17 -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/ubench/host_task_bench.cpp \
//       -Lgtn_amd/lib -lgtn_amd -L/opt/rocm/lib -lamdhip64 -pthread -o tools/ubench/host_task_bench
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "gtn/gtn.h"
using namespace gtn;
using Clock = std::chrono::steady_clock;
static double us(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
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
int main() {
  const int T = 1000, U = 100, M = 256, R = 512;
  std::mt19937 rng(7);
  std::vector<int> tg;
  for (int u = 0; u < U; ++u) tg.push_back(1 + rng() % (M - 1));
  float* dev = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&dev), sizeof(float) * size_t(R) * T * M) != hipSuccess) return 2;
  (void)scalarGraph(1.0f).item();  // runtime up
  for (int rep = 0; rep < 3; ++rep) {
    gtnx_parallel_enter();
    {
      auto a = Clock::now();
      for (int i = 0; i < R; ++i) { auto g = ctcGraph(tg); }
      auto b = Clock::now();
      std::printf("ctcGraph %.2f us | ", us(a, b) / R);
    }
    {
      auto a = Clock::now();
      for (int i = 0; i < R; ++i) { auto g = linearGraph(T, M); }
      auto b = Clock::now();
      std::printf("linearGraph %.2f us | ", us(a, b) / R);
    }
    {
      auto a = Clock::now();
      for (int i = 0; i < R; ++i) { auto g = linearGraph(T, M); g.setWeights(dev + size_t(i) * T * M); }
      auto b = Clock::now();
      std::printf("linearGraph+setWeights(device) %.2f us | ", us(a, b) / R);
    }
    {
      std::vector<Graph> keep;
      keep.reserve(R);
      auto a = Clock::now();
      for (int i = 0; i < R; ++i) {
        auto ctc = ctcGraph(tg);
        auto em = linearGraph(T, M);
        em.setWeights(dev + size_t(i) * T * M);
        keep.push_back(subtract(forwardScore(em), forwardScore(intersect(ctc, em))));
      }
      auto b = Clock::now();
      std::printf("whole fwd lambda %.2f us\n", us(a, b) / R);
      gtnx_parallel_leave();
      a = Clock::now();
      gtnx_parallel_flush();
      b = Clock::now();
      std::printf("  flush of %d tasks %.1f us", R, us(a, b));
      a = Clock::now();
      keep.clear();
      gtnx_reclaim();
      b = Clock::now();
      std::printf("  release + reclaim %.1f us\n", us(a, b));
    }
  }
  // pool wake-up + join: GTN_AMD_THREADS (read once per process) sets the pool size, default n / 16 threads
  for (int n : {64, 512}) {
    std::vector<int> v(n, 1);
    double best = 1e30, sum = 0;
    for (int it = 0; it < 300; ++it) {
      auto a = Clock::now();
      parallelMap([](int x) { return x + 1; }, v);
      auto b = Clock::now();
      if (it >= 50) {
        best = std::min(best, us(a, b));
        sum += us(a, b);
      }
    }
    const char* e = std::getenv("GTN_AMD_THREADS");
    std::printf("empty parallelMap over %4d ints (GTN_AMD_THREADS=%s): mean %.1f us, best %.1f us\n", n, e ? e : "default", sum / 250, best);
  }
  (void)hipFree(dev);
  return 0;
}

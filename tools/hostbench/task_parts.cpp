// task_parts.cpp -- where the host time of ONE task of the reference's CTC loop goes (benchmarks/ctc.cpp:150-160:
// ctcGraph, linearGraph, setWeights, the four graph-function calls) when 1 .. N threads run such tasks at once
// inside a parallelMap region.  Diagnostic for the GPU box (256 hardware threads): contention inside the tasks
// shows as per-task time growing with the thread count.
//   task_parts [threads=32] [tasks-per-thread=64]
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "gtn/gtn.h"
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
  const int NT = argc > 1 ? std::atoi(argv[1]) : 32, PER = argc > 2 ? std::atoi(argv[2]) : 64;
  const int M = 256, T = 1000, U = 100;
  float* dev = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&dev), sizeof(float) * size_t(64) * T * M) != hipSuccess) return 2;
  std::vector<int> target(U);
  for (int u = 0; u < U; ++u) target[u] = 1 + (u * 37) % (M - 1);
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  auto now = [] { return std::chrono::steady_clock::now(); };
  for (int nt : {1, 4, 8, 16, NT}) {
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    std::vector<double> part(size_t(nt) * 6, 0.0);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
      th.emplace_back([&, t] {
        double* p = &part[size_t(t) * 6];
        ready.fetch_add(1);
        while (!go.load()) {
        }
        for (int rep = 0; rep < 2; ++rep) {  // (the first repetition warms the thread's allocator arena)
          for (int k = 0; k < 6; ++k) p[k] = 0;
          gtnx_parallel_enter();
          {
            std::vector<Graph> keep;
            keep.reserve(size_t(PER));
            for (int i = 0; i < PER; ++i) {
              auto a0 = now();
              auto ctc = ctcGraph(target);
              auto a1 = now();
              auto em = linearGraph(T, M);
              auto a2 = now();
              em.setWeights(dev + size_t((t + i) % 64) * T * M);
              auto a3 = now();
              auto in = intersect(ctc, em);
              auto f1 = forwardScore(in);
              auto f2 = forwardScore(em);
              auto a4 = now();
              keep.push_back(subtract(f2, f1));
              auto a5 = now();
              p[0] += us(a0, a1), p[1] += us(a1, a2), p[2] += us(a2, a3), p[3] += us(a3, a4), p[4] += us(a4, a5);
            }
            auto a6 = now();
            gtnx_parallel_leave();
            p[5] += us(a6, now());
            gtnx_parallel_flush();  // (every thread joins its own slice: the engine serialises them)
          }
        }
      });
    while (ready.load() < nt) {
    }
    go.store(true);
    for (auto& x : th) x.join();
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int t = 0; t < nt; ++t)
      for (int k = 0; k < 6; ++k) s[k] += part[size_t(t) * 6 + k];
    const double n = double(nt) * PER;
    std::printf("threads %3d: per task us: ctcGraph %.2f  linearGraph %.2f  setWeights %.2f  intersect+2 forwardScore %.2f  subtract %.2f | leave %.1f us/thread\n",
                nt, s[0] / n, s[1] / n, s[2] / n, s[3] / n, s[4] / n, s[5] / nt);
  }
  (void)hipFree(dev);
  return 0;
}

// small_step.cpp -- the HOST side of BASELINE configs C1 (one utterance through the per-graph functions) and C2
// (forwardScore of 256 linear chains through the vector overloads) against the null HIP device: host microseconds per
// repetition and, with GTN_HOST_SAMPLE=<file>, stack samples.  Diagnostic only.
//   make -C tools/nullhip && LD_PRELOAD=tools/nullhip/_bin/libnullhip.so tools/nullhip/_bin/small_step c1|c2 [iters]
#include <hip/hip_runtime_api.h>
#include <sys/resource.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
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
  const bool c2b = argc > 1 && !std::strcmp(argv[1], "c2b");  // C2 through gtn::Batch (batch records)
  const bool c2 = c2b || (argc > 1 && !std::strcmp(argv[1], "c2"));
  const int iters = argc > 2 ? atoi(argv[2]) : 2000;
  const char* sample = std::getenv("GTN_HOST_SAMPLE");
  const int B = c2 ? 256 : 1, T = c2 ? 150 : 100, C = c2 ? 32 : 28, U = 20;
  float* dev = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&dev), sizeof(float) * size_t(B) * T * C) != hipSuccess) return 2;
  float* out = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&out), sizeof(float) * size_t(B)) != hipSuccess) return 2;
  std::vector<int> tg;
  for (int u = 0; u < U; ++u) tg.push_back(1 + (u * 7) % (C - 1));
  std::vector<gtnx_graph_t> hs(static_cast<size_t>(B));
  auto once = [&]() {
    if (c2b) {
      Batch ems = Batch::linear(B, T, C, dev, /*calcGrad=*/true, /*borrow=*/true);
      batched::forwardScore(ems).itemsToDevice(out);
    } else if (c2) {
      auto ems = linearGraphs(B, T, C, dev);
      auto scores = batched::forwardScore(ems);
      for (int b = 0; b < B; ++b) hs[size_t(b)] = scores[size_t(b)].handle();
      detail::check(gtnx_items_device_n(hs.data(), B, out));
    } else {
      Graph ctc = ctcGraph(tg);
      Graph em = linearGraph(T, C);
      em.setWeights(dev);
      Graph loss = subtract(forwardScore(em), forwardScore(intersect(ctc, em)));
      backward(loss);
      (void)loss.item();
    }
  };
  for (int i = 0; i < 50; ++i) once();
  if (sample) start_sampler();
  struct rusage r0, r1;
  getrusage(RUSAGE_SELF, &r0);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) once();
  getrusage(RUSAGE_SELF, &r1);
  std::printf("minor page faults per repetition: %.1f\n", double(r1.ru_minflt - r0.ru_minflt) / iters);
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
  if (sample) dump_samples(sample);
  std::printf("%s host us per repetition: %.2f\n", c2b ? "C2 (gtn::Batch)" : c2 ? "C2" : "C1", us);
  return 0;
}

// bm_ctc_c256.cpp -- timeBatchedCtc of the reference's benchmarks/ctc.cpp:136-168 at BASELINE config C3
// (T = 1000, U = 100, alphabet 256, B = 512), written with the reference's names only: ctcGraph built with
// addNode / addArc / arcSort, linearGraph + setWeights per utterance, parallelMap(fwd) then parallelMap(bwd).
// The one difference to the reference program is where the emissions live: in DEVICE memory (the caller of a
// GPU criterion has them there -- pytorch_loss.py:46-71 --, and setWeights takes a device address on this
// engine); `host` as 4th argument keeps them in host vectors exactly like the reference (then a step carries
// B*T*M*4 bytes over PCIe).
//   bm_ctc_c256 [B=512] [M=256] [iters=20] [device|host] [check]
// `check`: after the timing, the losses and emission gradients of a parallelMap run are compared with the same
// functions called one utterance at a time outside any parallelMap (exit code 1 on a mismatch): once with the
// one-at-a-time lattices kept symbolic (gtn::SymbolicCompose -- the same sweep kernels, so the two must agree
// to 1e-5: this checks the deferred execution itself) and once with them BUILT like the reference builds
// them (the float32 lattice recursion carries 8*eps*|score| ~ 4e-3 of rounding at T = 1000, DESIGN.md
// section 4, so that comparison allows 1e-2).
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
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
    if (l > 0) {
      ctc.addArc(l - 1, l, label);
    }
    if (l % 2 && l > 1 && label != target[idx - 1]) {
      ctc.addArc(l - 2, l, label);
    }
  }
  ctc.arcSort();
  return ctc;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? std::atoi(argv[1]) : 512;
  const int M = argc > 2 ? std::atoi(argv[2]) : 256;
  const int iters = argc > 3 ? std::atoi(argv[3]) : 20;
  const bool onDevice = !(argc > 4 && std::string(argv[4]) == "host");
  const bool check = argc > 5 && std::string(argv[5]) == "check";
  const int T = 1000, U = 100;

  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> uni(-5.f, 5.f);
  std::vector<std::vector<int>> targets(B);
  std::vector<std::vector<float>> hostScores(B);
  for (int b = 0; b < B; ++b) {
    for (int u = 0; u < U; ++u) targets[b].push_back(1 + int(rng() % unsigned(M - 1)));
    hostScores[b].resize(size_t(T) * M);
    for (auto& v : hostScores[b]) v = uni(rng);
  }
  // per-utterance emission buffers (rows of one device tensor, or the host vectors)
  std::vector<const float*> scores(B);
  float* dev = nullptr;
  if (onDevice) {
    if (hipMalloc(reinterpret_cast<void**>(&dev), sizeof(float) * size_t(B) * T * M) != hipSuccess) {
      std::fprintf(stderr, "hipMalloc failed\n");
      return 2;
    }
    for (int b = 0; b < B; ++b) {
      (void)hipMemcpy(dev + size_t(b) * T * M, hostScores[b].data(), sizeof(float) * size_t(T) * M, hipMemcpyHostToDevice);
      scores[b] = dev + size_t(b) * T * M;
    }
  } else {
    for (int b = 0; b < B; ++b) scores[b] = hostScores[b].data();
  }

  auto fwd = [T, M](const std::vector<int>& target, const float* emissionsScore) {
    auto ctc = ctcGraph(target);
    auto emissions = linearGraph(T, M);
    emissions.setWeights(emissionsScore);
    return subtract(forwardScore(emissions), forwardScore(intersect(ctc, emissions)));
  };
  auto bwd = [](const Graph& g) { backward(g); };

  float last = 0;
  const bool phases = std::getenv("BM_PHASES") != nullptr;  // host wall time per phase (diagnostic)
  double ph[4] = {0, 0, 0, 0};
  auto ctcBatched = [&]() {
    const auto a0 = std::chrono::steady_clock::now();
    {
      auto lossGraphs = parallelMap(fwd, targets, scores);
      const auto a1 = std::chrono::steady_clock::now();
      parallelMap(bwd, lossGraphs);
      const auto a2 = std::chrono::steady_clock::now();
      last = lossGraphs.back().item();  // (the engine runs asynchronously: the step ends when a result is read)
      const auto a3 = std::chrono::steady_clock::now();
      ph[0] += std::chrono::duration<double, std::milli>(a1 - a0).count();
      ph[1] += std::chrono::duration<double, std::milli>(a2 - a1).count();
      ph[2] += std::chrono::duration<double, std::milli>(a3 - a2).count();
    }
    ph[3] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a0).count();
  };

  for (int i = 0; i < 5; ++i) ctcBatched();
  (void)hipDeviceSynchronize();
  ph[0] = ph[1] = ph[2] = ph[3] = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) ctcBatched();
  // (item() returns when the LOSS has been computed; the step's backward sweep may still be running while the
  //  host prepares the next step -- the clock stops when the device has finished everything)
  (void)hipDeviceSynchronize();
  const auto t1 = std::chrono::steady_clock::now();
  const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count() / iters;
  std::printf("{\"program\": \"bm_ctc_c256\", \"B\": %d, \"T\": %d, \"U\": %d, \"alphabet\": %d, \"emissions\": \"%s\", "
              "\"ctcBatched_ms\": %.4f, \"losses_per_s\": %.1f, \"last_loss\": %.4f}\n",
              B, T, U, M, onDevice ? "device" : "host", ms, 1e3 * B / ms, last);

  if (phases)
    std::fprintf(stderr, "host ms per step: parallelMap(fwd) %.3f  parallelMap(bwd) %.3f  item %.3f  (step incl. release %.3f)\n",
                 ph[0] / iters, ph[1] / iters, ph[2] / iters, ph[3] / iters);
  int rc = 0;
  if (check) {
    // the same functions, one utterance at a time, no parallelMap
    struct Out {
      Graph loss, em;
    };
    (void)0;
    auto fwd2 = [T, M](const std::vector<int>& target, const float* emissionsScore) {
      auto ctc = ctcGraph(target);
      auto emissions = linearGraph(T, M);
      emissions.setWeights(emissionsScore);
      return Out{subtract(forwardScore(emissions), forwardScore(intersect(ctc, emissions))), emissions};
    };
    std::vector<Out> outs = parallelMap(fwd2, targets, scores);
    std::vector<Graph> lossGraphs;
    for (auto& o : outs) lossGraphs.push_back(o.loss);
    // setWeights COPIES (graph.cpp:179-181): once parallelMap has returned the caller may do what it likes with its
    // buffer.  The engine makes that copy inside the forward sweep of the region, so: wipe
    // the emissions of the utterances that are checked below before the backward pass runs, and put them back after
    const int nCheck = std::min(B, 24);
    if (onDevice) {
      (void)hipDeviceSynchronize();
      (void)hipMemset(dev, 0, sizeof(float) * size_t(nCheck) * T * M);
      (void)hipDeviceSynchronize();
    }
    parallelMap(bwd, lossGraphs);
    for (int b = 0; b < nCheck && onDevice; ++b) {  // the graphs still hold the values they were given
      const float* w = outs[b].em.weights();
      bool same = true;
      for (size_t i = 0; i < size_t(T) * M && same; ++i) same = w[i] == hostScores[b][i];
      if (!same) {
        std::printf("{\"check\": \"setWeights value semantics\", \"utterance\": %d, \"ok\": false}\n", b);
        rc = 1;
      }
    }
    if (onDevice) {
      (void)hipDeviceSynchronize();
      for (int b = 0; b < nCheck; ++b)
        (void)hipMemcpy(dev + size_t(b) * T * M, hostScores[b].data(), sizeof(float) * size_t(T) * M, hipMemcpyHostToDevice);
    }
    for (int built = 0; built < 2; ++built) {
      double worstLoss = 0, worstGrad = 0;
      for (int b = 0; b < nCheck; ++b) {
        Out ref;
        if (built) {
          ref = fwd2(targets[b], scores[b]);
        } else {
          SymbolicCompose symbolic;
          ref = fwd2(targets[b], scores[b]);
        }
        backward(ref.loss);
        const float l0 = ref.loss.item(), l1 = outs[b].loss.item();
        worstLoss = std::max(worstLoss, double(std::fabs(l0 - l1)) / std::max(1.0, double(std::fabs(l0))));
        Graph g0 = ref.em.grad(), g1 = outs[b].em.grad();
        const float* w0 = g0.weights();
        const float* w1 = g1.weights();
        for (size_t i = 0; i < size_t(T) * M; ++i) worstGrad = std::max(worstGrad, double(std::fabs(w0[i] - w1[i])));
      }
      std::printf("{\"check\": \"parallelMap vs one at a time (%s lattices)\", \"utterances\": %d, \"worst_rel_loss\": %.3g, "
                  "\"worst_abs_grad\": %.3g}\n", built ? "built" : "symbolic", nCheck, worstLoss, worstGrad);
      const double tol = built ? 1e-2 : 1e-5;
      if (!(worstLoss <= 1e-5) || !(worstGrad <= tol)) rc = 1;
    }
  }
  if (dev) (void)hipFree(dev);
  return rc;
}

// region_test.cpp -- the semantics of the per-graph functions INSIDE gtn::parallelMap on this engine: the calls
// are deferred to the region's join (include/gtn_amd.h: gtnx_parallel_enter / gtnx_parallel_flush) and must
// still behave like the reference's immediate calls (gtn/parallel/parallel_map.h:153-188 over
// gtn/functions.cpp): results looked at inside the task, graphs changed after a call, errors, backward twice,
// broadcast inputs, copies of results, non-uniform tasks, targets changed after arcSort, the calling thread's
// compose mode, backward of a part of a region's results.
// Own test program (not reference code); built by tests/dropin/Makefile, run by tests/test_dropin_gpu.py.
#include <cmath>
#include <cstdio>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "gtn/gtn.h"

using namespace gtn;

static int failures = 0;
#define EXPECT(cond)                                                     \
  do {                                                                   \
    if (!(cond)) {                                                       \
      ++failures;                                                        \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
    }                                                                    \
  } while (0)

static bool close(float a, float b, float tol = 1e-4f) { return std::fabs(a - b) <= tol * std::fmax(1.f, std::fabs(b)); }

static Graph ctcGraph(const std::vector<int>& target) {
  Graph ctc;
  const int L = 2 * (int)target.size() + 1;
  for (int l = 0; l < L; l++) {
    const int idx = (l - 1) / 2;
    ctc.addNode(l == 0, l == L - 1 || l == L - 2);
    const int label = l % 2 ? target[idx] : 0;
    ctc.addArc(l, l, label);
    if (l > 0) ctc.addArc(l - 1, l, label);
    if (l % 2 && l > 1 && label != target[idx - 1]) ctc.addArc(l - 2, l, label);
  }
  ctc.arcSort();
  return ctc;
}

int main() {
  const int B = 48, T = 40, M = 12;
  std::mt19937 rng(11);
  std::uniform_real_distribution<float> ud(-3.f, 3.f);
  std::vector<std::vector<int>> targets(B);
  std::vector<std::vector<float>> scores(B), scores2(B);
  for (int b = 0; b < B; ++b) {
    const int U = 1 + int(rng() % 8);
    for (int u = 0; u < U; ++u) targets[b].push_back(1 + int(rng() % (M - 1)));
    scores[b].resize(size_t(T) * M);
    scores2[b].resize(size_t(T) * M);
    for (auto& v : scores[b]) v = ud(rng);
    for (auto& v : scores2[b]) v = ud(rng);
  }
  std::vector<int> idx(B);
  for (int b = 0; b < B; ++b) idx[b] = b;

  // ---- 1. a result looked at inside its own task (sizes, item) is there at that moment
  {
    auto task = [&](int b) {
      Graph em = linearGraph(T, M);
      em.setWeights(scores[b].data());
      Graph lat = intersect(ctcGraph(targets[b]), em);
      Graph s = forwardScore(lat);
      const float inTask = s.item();
      const size_t arcs = lat.numArcs();  // builds this one lattice
      return std::make_pair(inTask, arcs);
    };
    auto got = parallelMap(task, idx);
    for (int b = 0; b < B; ++b) {
      Graph em = linearGraph(T, M);
      em.setWeights(scores[b].data());
      Graph lat = intersect(ctcGraph(targets[b]), em);
      EXPECT(close(got[b].first, forwardScore(lat).item()));
      EXPECT(got[b].second == lat.numArcs());
    }
  }

  // ---- 2. a graph changed AFTER a call keeps the call's view of it (setWeights between two forwardScores)
  {
    auto task = [&](int b) {
      Graph em = linearGraph(T, M);
      em.setWeights(scores[b].data());
      Graph f1 = forwardScore(em);
      em.setWeights(scores2[b].data());
      Graph f2 = forwardScore(em);
      return std::vector<Graph>{f1, f2};
    };
    auto got = parallelMap(task, idx);
    for (int b = 0; b < B; ++b) {
      Graph em = linearGraph(T, M);
      em.setWeights(scores[b].data());
      const float w1 = forwardScore(em).item();
      em.setWeights(scores2[b].data());
      const float w2 = forwardScore(em).item();
      EXPECT(close(got[b][0].item(), w1));
      EXPECT(close(got[b][1].item(), w2));
      EXPECT(!close(w1, w2, 1e-6f));
    }
  }

  // ---- 3. an error of one task comes out of parallelMap with the function's own type and message
  {
    auto task = [&](int b) {
      Graph g;
      g.addNode(true);
      g.addNode(false, true);
      g.addArc(0, 1, 0);
      if (b == 17) g.addArc(1, 1, 0);  // self-loop: forwardScore must throw (functions_test.cpp:238-250)
      return forwardScore(g);
    };
    bool threw = false;
    try {
      parallelMap(task, idx);
    } catch (const std::invalid_argument& e) {
      threw = std::string(e.what()).find("cycle") != std::string::npos;
    }
    EXPECT(threw);
    // ... and the engine is fine afterwards
    auto ok = parallelMap([&](int) { return negate(scalarGraph(2.0f)); }, idx);
    EXPECT(close(ok[3].item(), -2.0f));
  }

  // ---- 4. backward twice without retain for ONE root: that call fails, the other roots' gradients are those
  //         of single calls (autograd.cpp:42-45; nobody's gradient is accumulated twice)
  {
    std::vector<Graph> ems(B), losses;
    auto fwd = [&](int b) {
      ems[b] = linearGraph(T, M);
      ems[b].setWeights(scores[b].data());
      // (negate on top so that the roots are per-graph results, not elements of one batch record)
      return b % 2 ? negate(forwardScore(ems[b])) : forwardScore(ems[b]);
    };
    losses = parallelMap(fwd, idx);
    backward(losses[5]);  // root 5 has been run already
    bool threw = false;
    try {
      parallelMap([](const Graph& g) { backward(g); }, losses);
    } catch (const std::invalid_argument& e) {
      threw = std::string(e.what()).find("Cannot Backward twice") != std::string::npos;
    }
    EXPECT(threw);
    for (int b = 0; b < B; ++b) {
      Graph em = linearGraph(T, M);
      em.setWeights(scores[b].data());
      Graph l = b % 2 ? negate(forwardScore(em)) : forwardScore(em);
      backward(l);
      const float* want = em.grad().weights();
      const float* have = ems[b].grad().weights();
      double worst = 0;
      for (size_t i = 0; i < size_t(T) * M; ++i) worst = std::fmax(worst, std::fabs(double(want[i]) - have[i]));
      EXPECT(worst < 1e-5);
    }
  }

  // ---- 5. broadcast input (one target for every utterance, parallel_map.h:77-89), copies of a result alias it
  {
    std::vector<Graph> ems(B);
    for (int b = 0; b < B; ++b) {
      ems[b] = linearGraph(T, M);
      ems[b].setWeights(scores[b].data());
    }
    std::vector<Graph> one{ctcGraph(targets[0])};
    auto task = [](const Graph& c, const Graph& e) {
      Graph s = forwardScore(intersect(c, e));
      Graph alias = s;  // (before the call has run)
      return std::vector<Graph>{s, alias};
    };
    auto got = parallelMap(task, one, ems);
    parallelMap([](const std::vector<Graph>& g) { backward(g[0]); }, got);
    Graph shared = ctcGraph(targets[0]);
    std::vector<float> acc(shared.numArcs(), 0.0f);
    for (int b = 0; b < B; ++b) {
      Graph s = forwardScore(intersect(shared, ems[b]));
      EXPECT(close(got[b][0].item(), s.item()));
      EXPECT(got[b][0].id() == got[b][1].id());
      EXPECT(close(got[b][1].item(), s.item()));
    }
    // the shared target's gradient is the sum over the utterances
    Graph ref = ctcGraph(targets[0]);
    for (int b = 0; b < B; ++b) {
      Graph em = linearGraph(T, M);
      em.setWeights(scores[b].data());
      backward(forwardScore(intersect(ref, em)));
    }
    const float* want = ref.grad().weights();
    const float* have = one[0].grad().weights();
    double worst = 0;
    for (size_t i = 0; i < ref.numArcs(); ++i)
      worst = std::fmax(worst, std::fabs(double(want[i]) - have[i]) / std::fmax(1.0, std::fabs(double(want[i]))));
    EXPECT(worst < 1e-4);
  }

  // ---- 6. tasks that do different things (no common pattern): still every task's own result
  {
    auto task = [&](int b) {
      Graph em = linearGraph(T, M);
      em.setWeights(scores[b].data());
      switch (b % 4) {
        case 0: return forwardScore(em);
        case 1: return viterbiScore(em);
        case 2: return add(forwardScore(em), viterbiScore(em));
        default: return subtract(forwardScore(em), forwardScore(intersect(ctcGraph(targets[b]), em)));
      }
    };
    auto got = parallelMap(task, idx);
    for (int b = 0; b < B; ++b) EXPECT(close(got[b].item(), task(b).item()));
  }

  // ---- 7. viterbiPath of a deferred composition, labels bit-exact
  {
    auto task = [&](int b) {
      Graph em = linearGraph(T, M);
      em.setWeights(scores[b].data());
      return viterbiPath(intersect(ctcGraph(targets[b]), em));
    };
    auto got = parallelMap(task, idx);
    for (int b = 0; b < B; ++b) EXPECT(got[b].labelsToVector() == task(b).labelsToVector());
  }

  // ---- 8. a CTC-shaped target changed by makeAccept AFTER arcSort is no longer the standard acceptor: the
  //         deferred product must see the extra accept node (graph.h:346-352; the shape cache of arcSort is stale)
  {
    auto target = [&](int b) {
      Graph c = ctcGraph(targets[b]);
      c.makeAccept(0);  // the empty alignment's start node also accepts
      return c;
    };
    auto task = [&](int b) {
      Graph em = linearGraph(T, M);
      em.setWeights(scores[b].data());
      return forwardScore(intersect(target(b), em));
    };
    auto got = parallelMap(task, idx);
    for (int b = 0; b < B; ++b) {
      // one at a time, lattice BUILT (the reference's own route through compose.cpp:377-522)
      SymbolicCompose built(0);
      Graph em = linearGraph(T, M);
      em.setWeights(scores[b].data());
      const float want = forwardScore(intersect(target(b), em)).item();
      Graph plain_em = linearGraph(T, M);
      plain_em.setWeights(scores[b].data());
      const float without = forwardScore(intersect(ctcGraph(targets[b]), plain_em)).item();
      EXPECT(close(got[b].item(), want));
      EXPECT(want >= without);  // (one more accepting state can only add paths)
    }
  }

  // ---- 9. gtnx_compose_mode of the calling thread goes with the tasks: under mode 0 a composition made inside
  //         parallelMap is BUILT -- its gradient exists after a retained backward (autograd_test.cpp:148-188 on
  //         compose results), which a symbolic product does not have
  {
    SymbolicCompose built(0);
    std::vector<Graph> lats(B);
    auto task = [&](int b) {
      Graph em = linearGraph(T, M);
      em.setWeights(scores[b].data());
      lats[b] = intersect(ctcGraph(targets[b]), em);
      Graph s = forwardScore(lats[b]);
      backward(s, true);
      return s;
    };
    auto got = parallelMap(task, idx);
    int with_grad = 0;
    for (int b = 0; b < B; ++b) {
      bool ok = false;
      try {
        ok = lats[b].grad().numArcs() == lats[b].numArcs();
      } catch (const std::logic_error&) {
        ok = false;
      }
      with_grad += ok;
    }
    EXPECT(with_grad == B);
  }

  // ---- 10. backward of SOME results of a region (every other one): only their inputs get a gradient
  {
    std::vector<Graph> ems(B);
    auto fwd = [&](int b) {
      ems[b] = linearGraph(T, M);
      ems[b].setWeights(scores[b].data());
      return subtract(forwardScore(ems[b]), forwardScore(intersect(ctcGraph(targets[b]), ems[b])));
    };
    auto losses = parallelMap(fwd, idx);
    std::vector<Graph> some;
    for (int b = 0; b < B; b += 2) some.push_back(losses[b]);
    parallelMap([](const Graph& g) { backward(g); }, some);
    for (int b = 0; b < B; ++b) {
      EXPECT(ems[b].isGradAvailable() == (b % 2 == 0));
      if (b % 2) continue;
      Graph em = linearGraph(T, M);
      em.setWeights(scores[b].data());
      Graph l = subtract(forwardScore(em), forwardScore(intersect(ctcGraph(targets[b]), em)));
      backward(l);
      const float* want = em.grad().weights();
      const float* have = ems[b].grad().weights();
      double worst = 0;
      for (size_t i = 0; i < size_t(T) * M; ++i) worst = std::fmax(worst, std::fabs(double(want[i]) - have[i]));
      EXPECT(worst < 1e-4);
      EXPECT(close(losses[b].item(), l.item()));
    }
  }

  if (failures) {
    std::printf("%d check(s) failed\n", failures);
    return 1;
  }
  std::printf("All tests passed\n");
  return 0;
}

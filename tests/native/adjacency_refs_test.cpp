// adjacency_refs_test.cpp -- the references include/gtn/graph.h hands out for out(n) / in(n) / start() / accept()
// stay valid while others are taken (the reference returns references into per-node vectors): nested loops over
// two nodes' lists, iterator pairs from two calls, and the four lists of one node at once.
// Own test program; built by tests/dropin/Makefile, run on the CPU over the reference-backed shim
// (tests/test_dropin_cpu.py) and on the engine (tests/test_dropin_gpu.py).
#include <algorithm>
#include <cstdio>
#include <vector>

#include "gtn/gtn.h"

using namespace gtn;

int main() {
  Graph g;
  const int N = 6;
  for (int n = 0; n < N; ++n) g.addNode(n < 2, n >= N - 2);
  // arc (s, d) for every s < d: node n has n in-arcs and N - 1 - n out-arcs
  std::vector<std::pair<int, int>> arcs;
  for (int s = 0; s < N; ++s)
    for (int d = s + 1; d < N; ++d) {
      g.addArc(s, d, s * N + d);
      arcs.push_back({s, d});
    }
  int bad = 0;
  // nested: the outer range must survive the inner calls
  for (int n = 0; n < N; ++n) {
    int seen = 0;
    for (int a : g.out(n)) {
      if (g.srcNode(a) != n) ++bad;
      int inner = 0;
      for (int b : g.out(g.dstNode(a))) inner += g.srcNode(b) == g.dstNode(a) ? 1 : 100;
      if (inner != N - 1 - g.dstNode(a)) ++bad;
      ++seen;
    }
    if (seen != N - 1 - n) ++bad;
  }
  // iterator pairs from two calls refer to the same storage
  for (int n = 0; n < N; ++n) {
    if (int(std::distance(g.in(n).begin(), g.in(n).end())) != n) ++bad;
    if (!std::is_sorted(g.out(n).begin(), g.out(n).end())) ++bad;
  }
  // four lists held at once
  const auto& st = g.start();
  const auto& ac = g.accept();
  const auto& o2 = g.out(2);
  const auto& i3 = g.in(3);
  const auto& o4 = g.out(4);
  if (st.size() != 2 || ac.size() != 2 || o2.size() != size_t(N - 3) || i3.size() != 3 || o4.size() != 1) ++bad;
  for (int a : o2)
    if (g.srcNode(a) != 2) ++bad;
  for (int a : i3)
    if (g.dstNode(a) != 3) ++bad;
  std::printf("%d inconsistencies\n", bad);
  if (bad) return 1;
  std::printf("All tests passed\n");
  return 0;
}

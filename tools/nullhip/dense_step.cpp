// dense_step.cpp -- forwardScore(compose(emissions, dense transitions)) + backward through the C++ shim against the
// null HIP device: the host side of the dense regime (lazy.hip's launch chains, the early beta sweep on the runtime's
// side stream and its enqueuing thread: GTNX_EAGER_BETA=1) as a sanitizer target.  Values are garbage by design.
//   LD_PRELOAD=tools/nullhip/_bin/libnullhip.so tools/nullhip/_bin/dense_step [steps] [B] [T] [C]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gtn/gtn.h"

using namespace gtn;

int main(int argc, char** argv) {
  const int steps = argc > 1 ? std::atoi(argv[1]) : 4, B = argc > 2 ? std::atoi(argv[2]) : 8;
  const int T = argc > 3 ? std::atoi(argv[3]) : 20, C = argc > 4 ? std::atoi(argv[4]) : 64;
  Graph trans;  // examples/asg.cpp: a start node and one node per label, every label reachable from every node
  trans.addNode(true, false);
  for (int c = 0; c < C; ++c) trans.addNode(false, true);
  for (int c = 0; c < C; ++c) trans.addArc(0, c + 1, c);
  for (int s = 0; s < C; ++s)
    for (int d = 0; d < C; ++d) trans.addArc(s + 1, d + 1, d);
  std::vector<float> em(size_t(T) * C, 0.25f);
  for (int s = 0; s < steps; ++s) {
    SymbolicCompose symbolic(1);
    std::vector<Graph> scores;
    for (int b = 0; b < B; ++b) {
      Graph e = linearGraph(T, C);
      e.setWeights(em.data());
      scores.push_back(forwardScore(compose(e, trans)));
    }
    if (s % 3 == 2) {  // dropped without a backward (the early sweep may still be enqueuing)
      std::printf("step %d: dropped\n", s);
      continue;
    }
    for (auto& sc : scores) backward(sc);
    std::printf("step %d: %d losses differentiated, transitions gradient has %d arcs\n", s, B, int(trans.grad().numArcs()));
    trans.zeroGrad();
  }
  // (the side stream's enqueuing thread exists once an early sweep ran: 2 threads with GTNX_EAGER_BETA=1, else 1)
  if (FILE* f = std::fopen("/proc/self/status", "r")) {
    char line[256];
    while (std::fgets(line, sizeof line, f))
      if (!std::strncmp(line, "Threads:", 8)) std::fputs(line, stdout);
    std::fclose(f);
  }
  return 0;
}

// asg_criterion.h -- the ASG criterion of examples/asg.cpp:30-68 (golden values in
// test/criterion_test.cpp:182-306), batched over utterances that share ONE transitions graph:
//   loss_b = forwardScore(emissions_b o transitions)                       (full-connect)
//          - forwardScore(emissions_b o (forceAlign_b o transitions))      (force-align)
// On this engine the full-connect product is never built (compose keeps it symbolic and the
// dense-regime kernels of lazy.hip run it; 262 M arcs per utterance at C=512, T=1000); the
// force-align acceptors composed with the transitions are built on the device as band records (batch.h).
// Header-only.
#pragma once

#include <cstdint>
#include <vector>

#include "gtn/gtn.h"

namespace gtn {
namespace criteria {

/** transitions over N labels: node 0 start, nodes 1..N accept; arcs 0 -> i+1 (label i), then
 *  j+1 -> i+1 (label i) for i, j in [0, N) in the order of examples/asg.cpp:36-47, i.e. arc
 *  N + i*N + j carries p(i | j).  Weights are left at 0. */
inline Graph asgTransitions(int N) {
  Graph g;
  g.addNode(true);
  for (int i = 1; i <= N; i++) {
    g.addNode(false, true);
    g.addArc(0, i, i - 1);
  }
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) g.addArc(j + 1, i + 1, i);
  g.arcSort();  // by ilabel: compose(forceAlign, transitions) then searches a node's arcs
  return g;
}

/** all alignments of `target` (examples/asg.cpp:50-57): U+1 nodes, a step and a self-loop per label */
inline Graph asgForceAlign(const std::vector<int>& target) {
  Graph fal(false);
  fal.addNode(true, target.empty());
  for (size_t l = 1; l <= target.size(); l++) {
    fal.addNode(false, l == target.size());
    fal.addArc(l - 1, l, target[l - 1]);
    fal.addArc(l, l, target[l - 1]);
  }
  return fal;
}

/** forward + backward for a batch.  `emissions`: device [B][T][N], read in place; `transitions`: the graph of
 *  asgTransitions(N) with the caller's weights (its gradient accumulates over the batch, as in
 *  criterion_test.cpp:289-305); `lossDev`: device [B]; `gradDev`: device [B][T][N] or null.
 *  Written on batch records (gtn/batch.h): the force-alignment acceptors composed with the transitions are
 *  built on the device from the label sequences (what took 137 of 154 ms per batch of 512 through the
 *  ordinary compose), the full-connect term runs through the per-graph functions on the batch's elements. */
inline void asgLossBatch(
    const void* emissions,
    const int* labels,
    const int* lengths,
    int B,
    int T,
    int N,
    Graph& transitions,
    void* lossDev,
    void* gradDev) {
  Batch ems = Batch::linear(B, T, N, emissions, gradDev != nullptr, /*borrow=*/true);
  std::vector<int64_t> off(B);
  for (int b = 0; b < B; ++b) off[b] = (int64_t)b * T * N;
  Batch trans(std::vector<Graph>{transitions});
  SymbolicCompose symbolic;  // the full-connect product (262 M arcs per utterance at C4) is never built
  Batch fcc = batched::forwardScore(batched::compose(ems, trans));
  Batch fals = Batch::asgForceAlign(labels, lengths, B, transitions, N);
  Batch fal = batched::forwardScore(batched::compose(ems, fals));
  Batch losses = batched::subtract(fcc, fal);
  if (gradDev || transitions.calcGrad()) batched::backward(losses);
  losses.itemsToDevice(lossDev);
  if (gradDev) ems.gradsToDevice(gradDev, off.data());
}

inline void asgLossBatch(
    const void* emissions,
    const std::vector<std::vector<int>>& targets,
    int T,
    int N,
    Graph& transitions,
    void* lossDev,
    void* gradDev) {
  std::vector<int> flat, len;
  for (auto& t : targets) {
    flat.insert(flat.end(), t.begin(), t.end());
    len.push_back((int)t.size());
  }
  asgLossBatch(emissions, flat.data(), len.data(), (int)targets.size(), T, N, transitions, lossDev, gradDev);
}

} // namespace criteria
} // namespace gtn

// gtn/rand.h -- reference gtn/rand.h:22-40 (test helpers; host-side over the public API)
#pragma once

#include <cmath>
#include <cstdlib>

#include "gtn/functions.h"
#include "gtn/utils.h"

namespace gtn {

/** Random accepting path of `g` (reference rand.cpp:14-74; uses rand() like it). */
inline Graph sample(const Graph& g, size_t maxLength = 1000) {
  if (!g.numStart() || !g.numAccept()) return Graph{};
  std::vector<int> arcs;
  size_t node = g.start()[std::rand() % g.numStart()];
  size_t acceptLength = 0;
  for (size_t length = 0; length < maxLength + 1; ++length) {
    const size_t choices = g.numOut(node) + (g.isAccept(node) ? 1 : 0);
    if (g.isAccept(node)) acceptLength = length + 1;
    if (!choices) return Graph{};  // dead end
    const size_t pick = std::rand() % choices;
    if (pick == g.numOut(node)) break;  // stop at this accepting node
    const int arc = g.out(node, pick);
    node = g.dstNode(arc);
    arcs.push_back(arc);
  }
  if (!acceptLength) return Graph{};
  arcs.resize(acceptLength - 1);
  auto gradFunc = [arcs](std::vector<Graph>& inputs, Graph& deltas) {
    if (!inputs[0].calcGrad()) return;
    std::vector<float> grad(inputs[0].numArcs(), 0.0f);
    for (size_t a = 0; a < deltas.numArcs(); ++a) grad[arcs[a]] += deltas.weight(a);
    inputs[0].addGrad(std::move(grad));
  };
  Graph path(gradFunc, {g});
  path.addNode(true, acceptLength == 1);
  for (size_t i = 1; i < acceptLength; ++i) {
    path.addNode(false, i + 1 == acceptLength);
    path.addArc(i - 1, i, g.ilabel(arcs[i - 1]), g.olabel(arcs[i - 1]), g.weight(arcs[i - 1]));
  }
  return path;
}

/** Monte-Carlo equivalence of two transducers (reference rand.cpp:77-126). */
inline bool randEquivalent(const Graph& g1, const Graph& g2, size_t numSamples = 100, double tol = 1e-4,
                           size_t maxLength = 1000) {
  for (size_t i = 0; i < numSamples; ++i) {
    Graph path = sample(std::rand() % 2 ? g1 : g2, maxLength);
    path.setCalcGrad(false);
    if (equal(path, Graph{})) continue;
    const Graph inp = projectInput(path), outp = projectOutput(path);
    auto restrict_to_path = [&](const Graph& g) {
      Graph c = compose(inp, g);
      c.setCalcGrad(false);
      return compose(c, outp);
    };
    const Graph c1 = restrict_to_path(g1), c2 = restrict_to_path(g2);
    const bool empty1 = equal(c1, Graph{}), empty2 = equal(c2, Graph{});
    if (empty1 != empty2) return false;
    if (empty1) continue;
    if (std::abs(forwardScore(c1).item() - forwardScore(c2).item()) > tol) return false;
  }
  return true;
}

} // namespace gtn

// gtn/rand.h -- sample / randEquivalent (declared at reference gtn/rand.h:22-40).
// Test utilities, host-side, written against the public Graph API of this package: the
// reference's functions_test.cpp checks its epsilon compositions with randEquivalent.
#pragma once

#include <cmath>
#include <cstdlib>
#include <optional>
#include <vector>

#include "gtn/functions.h"
#include "gtn/utils.h"

namespace gtn {

namespace detail {
/** A random walk over `g` from a random start node.  At a node every out-arc is as likely as
 *  every other and, on an accepting node, as stopping; the walk takes at most `maxArcs` + 1
 *  decisions.  Returns the arcs up to the LAST accepting node it stood on (nullopt when it never
 *  stood on one, or ran into a node with nothing to choose from). */
inline std::optional<std::vector<int>> walkToAccept(const Graph& g, size_t maxArcs) {
  if (g.numStart() == 0 || g.numAccept() == 0) return std::nullopt;
  std::vector<int> trail;
  std::optional<size_t> lastAccept;  // trail length when the walk last stood on an accepting node
  size_t here = g.start()[size_t(std::rand()) % g.numStart()];
  for (size_t decisions = 0; decisions <= maxArcs; ++decisions) {
    const bool canStop = g.isAccept(here);
    if (canStop) lastAccept = trail.size();
    const size_t fanout = g.numOut(here);
    const size_t options = fanout + (canStop ? 1 : 0);
    if (options == 0) return std::nullopt;
    const size_t choice = size_t(std::rand()) % options;
    if (choice == fanout) break;  // the extra option of an accepting node: stop here
    trail.push_back(g.out(here, choice));
    here = g.dstNode(trail.back());
  }
  if (!lastAccept) return std::nullopt;
  trail.resize(*lastAccept);
  return trail;
}

/** log-sum of the paths of `g` that read `inp` and write `outp`, if there is one. */
inline std::optional<float> scoreOfStrings(const Graph& inp, const Graph& g, const Graph& outp) {
  Graph left = compose(inp, g);
  left.setCalcGrad(false);
  Graph both = compose(left, outp);
  if (equal(both, Graph{})) return std::nullopt;
  return forwardScore(both).item();
}
} // namespace detail

/** A random accepting path of `g` as a chain graph; the empty graph if the walk found none. */
inline Graph sample(const Graph& g, size_t maxLength = 1000) {
  const auto trail = detail::walkToAccept(g, maxLength);
  if (!trail) return Graph{};
  const std::vector<int> arcs = *trail;
  // d path / d g: every arc of the chain is one arc of g
  Graph path(
      [arcs](std::vector<Graph>& inputs, Graph& deltas) {
        Graph& src = inputs[0];
        if (!src.calcGrad()) return;
        std::vector<float> grad(src.numArcs(), 0.0f);
        for (size_t k = 0; k < arcs.size(); ++k) grad[arcs[k]] += deltas.weight(k);
        src.addGrad(std::move(grad));
      },
      {g});
  const size_t n = arcs.size();
  for (size_t k = 0; k <= n; ++k) path.addNode(k == 0, k == n);
  for (size_t k = 0; k < n; ++k) path.addArc(k, k + 1, g.ilabel(arcs[k]), g.olabel(arcs[k]), g.weight(arcs[k]));
  return path;
}

/** Monte-Carlo test that two transducers assign the same score to the same string pairs:
 *  `numSamples` paths drawn from either graph, each scored in both. */
inline bool randEquivalent(const Graph& g1, const Graph& g2, size_t numSamples = 100, double tol = 1e-4,
                           size_t maxLength = 1000) {
  for (size_t trial = 0; trial < numSamples; ++trial) {
    const Graph& from = (std::rand() & 1) ? g1 : g2;
    Graph path = sample(from, maxLength);
    path.setCalcGrad(false);
    if (equal(path, Graph{})) continue;  // nothing sampled this time
    const Graph inp = projectInput(path), outp = projectOutput(path);
    const auto s1 = detail::scoreOfStrings(inp, g1, outp), s2 = detail::scoreOfStrings(inp, g2, outp);
    if (s1.has_value() != s2.has_value()) return false;  // one of them does not accept the pair at all
    if (s1 && std::abs(*s1 - *s2) > tol) return false;
  }
  return true;
}

} // namespace gtn

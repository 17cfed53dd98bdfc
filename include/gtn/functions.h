// gtn/functions.h -- reference gtn/functions.h:19-152 for the hot path, plus the
// batched (std::vector<Graph>) overloads the reference exposes through its Python
// binding (bindings/python/gtn/_functions.cpp:84-135).  A vector call is ONE
// batched device launch per kernel family.
#pragma once

#include <algorithm>
#include <vector>

#include "gtn/graph.h"

namespace gtn {
namespace detail {
inline std::vector<gtnx_graph_t> handles(const std::vector<Graph>& v) {
  std::vector<gtnx_graph_t> h;
  h.reserve(v.size());
  for (auto& g : v) h.push_back(g.handle());
  return h;
}
inline std::vector<Graph> adopt(std::vector<gtnx_graph_t>& h) {
  std::vector<Graph> out;
  out.reserve(h.size());
  for (auto x : h) out.push_back(Graph::fromHandle(x));
  return out;
}
template <class F>
Graph unary(F f, const Graph& g) {
  gtnx_graph_t out;
  check(f(g.handle(), &out));
  return Graph::fromHandle(out);
}
template <class F>
Graph binary(F f, const Graph& a, const Graph& b) {
  gtnx_graph_t out;
  check(f(a.handle(), b.handle(), &out));
  return Graph::fromHandle(out);
}
template <class F>
std::vector<Graph> unaryN(F f, const std::vector<Graph>& g) {
  auto h = handles(g);
  std::vector<gtnx_graph_t> out(h.size());
  if (!h.empty()) check(f(h.data(), (int)h.size(), out.data()));
  return adopt(out);
}
template <class F>
std::vector<Graph> binaryN(F f, const std::vector<Graph>& a, const std::vector<Graph>& b) {
  auto ha = handles(a), hb = handles(b);
  std::vector<gtnx_graph_t> out(std::max(ha.size(), hb.size()));
  if (!out.empty()) check(f(ha.data(), (int)ha.size(), hb.data(), (int)hb.size(), out.data()));
  return adopt(out);
}
} // namespace detail

inline Graph negate(const Graph& g) { return detail::unary(&gtnx_negate, g); }
inline Graph add(const Graph& g1, const Graph& g2) { return detail::binary(&gtnx_add, g1, g2); }
inline Graph subtract(const Graph& g1, const Graph& g2) { return detail::binary(&gtnx_subtract, g1, g2); }
inline Graph compose(const Graph& g1, const Graph& g2) { return detail::binary(&gtnx_compose, g1, g2); }
inline Graph intersect(const Graph& g1, const Graph& g2) { return detail::binary(&gtnx_intersect, g1, g2); }
inline Graph forwardScore(const Graph& g) { return detail::unary(&gtnx_forward_score, g); }
inline Graph viterbiScore(const Graph& g) { return detail::unary(&gtnx_viterbi_score, g); }
inline Graph viterbiPath(const Graph& g) { return detail::unary(&gtnx_viterbi_path, g); }

/** Engine extension (not in the reference): while alive, compose / intersect on this thread run under the
 *  given gtnx_compose_mode (gtn_amd.h).  Nobody has to use it: by default (-1) the product of an emissions
 *  chain and a small epsilon-free graph built on the host already stays symbolic -- forwardScore /
 *  viterbiScore / viterbiPath sweep it without building it.  SymbolicCompose(2) extends that to partners
 *  that only exist on the device, (1) to every eligible partner, (0) builds every composition (a caller
 *  that reads the composition's own grad() after a retained backward wants that). */
class SymbolicCompose {
 public:
  explicit SymbolicCompose(int mode = 2) { detail::check(gtnx_compose_mode(mode, &prev_)); }
  ~SymbolicCompose() { gtnx_compose_mode(prev_, nullptr); }
  SymbolicCompose(const SymbolicCompose&) = delete;
  SymbolicCompose& operator=(const SymbolicCompose&) = delete;

 private:
  int prev_ = 0;
};

// Batched forms live in gtn::batched so that the plain names stay un-overloaded
// (reference callers pass them as function pointers, e.g. parallelMap(negate, v)).
namespace batched {
inline std::vector<Graph> negate(const std::vector<Graph>& g) { return detail::unaryN(&gtnx_negate_n, g); }
inline std::vector<Graph> add(const std::vector<Graph>& a, const std::vector<Graph>& b) {
  return detail::binaryN(&gtnx_add_n, a, b);
}
inline std::vector<Graph> subtract(const std::vector<Graph>& a, const std::vector<Graph>& b) {
  return detail::binaryN(&gtnx_subtract_n, a, b);
}
inline std::vector<Graph> compose(const std::vector<Graph>& a, const std::vector<Graph>& b) {
  return detail::binaryN(&gtnx_compose_n, a, b);
}
inline std::vector<Graph> intersect(const std::vector<Graph>& a, const std::vector<Graph>& b) {
  return detail::binaryN(&gtnx_intersect_n, a, b);
}
inline std::vector<Graph> forwardScore(const std::vector<Graph>& g) { return detail::unaryN(&gtnx_forward_score_n, g); }
inline std::vector<Graph> viterbiScore(const std::vector<Graph>& g) { return detail::unaryN(&gtnx_viterbi_score_n, g); }
inline std::vector<Graph> viterbiPath(const std::vector<Graph>& g) { return detail::unaryN(&gtnx_viterbi_path_n, g); }
} // namespace batched

// ---------------------------------------------------------------------------
// Rational / structural operations (reference gtn/functions.h:45-123).  clone, the projections,
// concat, closure and union_ are built by the engine on the device (gtnx_clone / gtnx_concat /
// gtnx_closure / gtnx_union: rational.hip), node and arc ids as functions.cpp:66-223 numbers them,
// gradients of the inputs as slices of the output's; remove (gtnx_remove) as well.
// ---------------------------------------------------------------------------
enum class Projection { NONE = 0, INPUT = 1, OUTPUT = 2 };

inline Graph clone(const Graph& g, Projection projection = Projection::NONE) {
  gtnx_graph_t h;
  detail::check(gtnx_clone(g.handle(), static_cast<int>(projection), &h));
  return Graph::fromHandle(h);
}
inline Graph projectInput(const Graph& g) { return clone(g, Projection::INPUT); }
inline Graph projectOutput(const Graph& g) { return clone(g, Projection::OUTPUT); }

inline Graph concat(const std::vector<Graph>& graphs) {
  auto hs = detail::handles(graphs);
  gtnx_graph_t h;
  detail::check(gtnx_concat(hs.data(), static_cast<int>(hs.size()), &h));
  return Graph::fromHandle(h);
}
inline Graph concat(const Graph& g1, const Graph& g2) { return concat(std::vector<Graph>{g1, g2}); }

inline Graph closure(const Graph& g) {
  gtnx_graph_t h;
  detail::check(gtnx_closure(g.handle(), &h));
  return Graph::fromHandle(h);
}

inline Graph union_(const std::vector<Graph>& graphs) {
  auto hs = detail::handles(graphs);
  gtnx_graph_t h;
  detail::check(gtnx_union(hs.data(), static_cast<int>(hs.size()), &h));
  return Graph::fromHandle(h);
}

inline Graph remove(const Graph& g, int ilabel, int olabel) {
  // functions.cpp:257-318: arcs labelled (ilabel:olabel) are contracted; no gradient.  Built by the engine on the
  // device (gtnx_remove: rational.hip), node and arc ids as the reference's breadth-first construction numbers them.
  gtnx_graph_t h;
  detail::check(gtnx_remove(g.handle(), ilabel, olabel, &h));
  return Graph::fromHandle(h);
}
inline Graph remove(const Graph& g, int label = epsilon) { return remove(g, label, label); }

} // namespace gtn

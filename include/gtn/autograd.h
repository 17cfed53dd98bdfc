// gtn/autograd.h -- reference gtn/autograd.h:27,37 (+ the batched form)
#pragma once
#include "gtn/functions.h"

namespace gtn {
inline void backward(Graph g, bool retainGraph = false) { detail::check(gtnx_backward(g.handle(), retainGraph)); }
inline void backward(Graph g, const Graph& grad, bool retainGraph = false) {
  detail::check(gtnx_backward_with_grad(g.handle(), grad.handle(), retainGraph));
}
namespace batched {
inline void backward(const std::vector<Graph>& graphs, bool retainGraph = false) {
  auto h = detail::handles(graphs);
  if (!h.empty()) detail::check(gtnx_backward_n(h.data(), (int)h.size(), retainGraph));
}
} // namespace batched
} // namespace gtn
